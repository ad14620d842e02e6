"""ComfyUI node layer for the MVs_Algorithms hot path: argument plumbing only.

Mirror of the eight nodes of /root/reference/nodes.py that sit on the path -- same class names (ComfyUI mangles them to
"[Comfy3D] " + name.replace("_", " "), __init__.py:53-63, so saved workflow JSON keeps resolving), the same INPUT_TYPES keys
and defaults, RETURN_TYPES / RETURN_NAMES / FUNCTION and tensor layouts:
  Load_3DGS :323-356   Save_3DGS :387-415   Switch_3DGS_Axis :676-705   Stack_Orbit_Camera_Poses :792-980
  Mesh_Orbit_Renderer :1011-1097   Gaussian_Splatting_Orbit_Renderer :1100-1163   Gaussian_Splatting_3D :1165-1313
  Fitting_Mesh_With_Multiview_Images :1315-1418
ComfyUI itself (folder_paths, ProgressBar) is optional: the classes run head-less.  Everything that computes lives in
MVs_Algorithms/* and, below that, in libc3d_hip.so."""
import math
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from c3d_hip.ply import PlyData
from mesh_processer.mesh_utils import calculate_max_sh_degree_from_gs_ply, switch_ply_axis_and_scale
from MVs_Algorithms.DiffRastMesh.diff_mesh import DiffMesh, DiffMeshCameraController
from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams, GaussianSplatting3D, GaussianSplattingCameraController
from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
from shared_utils.camera_utils import AZIMUTH_MAX, AZIMUTH_MIN, ELEVATION_MAX, ELEVATION_MIN

try:   # inside ComfyUI
    import folder_paths as comfy_paths
    _IN, _OUT = comfy_paths.input_directory, comfy_paths.output_directory
except Exception:   # head-less
    _IN = _OUT = os.getcwd()

SUPPORTED_3DGS_EXTENSIONS = ('.ply',)
_POSES = ("ORBIT_CAMPOSES",)
_F = lambda d, **k: ("FLOAT", dict(default=d, **k))
_I = lambda d, **k: ("INT", dict(default=d, **k))
_BIG = 0xffffffffffffffff


def _warn(node, msg):
    print("[%s] %s" % (node.__class__.__name__, msg), file=sys.stderr)


class Load_3DGS:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"gs_file_path": ("STRING", {"default": '', "multiline": False})}}
    RETURN_TYPES, RETURN_NAMES, FUNCTION, CATEGORY = ("GS_PLY",), ("gs_ply",), "load_gs", "Comfy3D/Import|Export"

    def load_gs(self, gs_file_path):
        path = gs_file_path if os.path.isabs(gs_file_path) else os.path.join(_IN, gs_file_path)
        if not os.path.exists(path):
            _warn(self, "File %s does not exist" % path)
            return (None,)
        if not path.lower().endswith(SUPPORTED_3DGS_EXTENSIONS):
            _warn(self, "File name %s does not end with supported 3DGS file extensions: %s" % (os.path.basename(path), SUPPORTED_3DGS_EXTENSIONS))
            return (None,)
        return (PlyData.read(path),)


class Save_3DGS:
    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {"gs_ply": ("GS_PLY",), "save_path": ("STRING", {"default": '3DGS_%Y-%m-%d-%M-%S-%f.ply', "multiline": False})}}
    OUTPUT_NODE = True
    RETURN_TYPES, RETURN_NAMES, FUNCTION, CATEGORY = ("STRING",), ("save_path",), "save_gs", "Comfy3D/Import|Export"

    def save_gs(self, gs_ply, save_path):
        import datetime
        save_path = datetime.datetime.now().strftime(save_path)
        if not os.path.isabs(save_path):
            save_path = os.path.join(_OUT, save_path)
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        if not save_path.lower().endswith(SUPPORTED_3DGS_EXTENSIONS):
            _warn(self, "File name %s does not end with supported 3DGS file extensions: %s" % (os.path.basename(save_path), SUPPORTED_3DGS_EXTENSIONS))
            return ("",)
        gs_ply.write(save_path)
        return (save_path,)


class Switch_3DGS_Axis:
    @classmethod
    def INPUT_TYPES(cls):
        ax = ["+x", "-x", "+y", "-y", "+z", "-z"]
        return {"required": {"gs_ply": ("GS_PLY",), "axis_x_to": (ax, {"default": "+x"}), "axis_y_to": (ax, {"default": "+y"}),
                             "axis_z_to": (ax, {"default": "+z"})}}
    RETURN_TYPES, RETURN_NAMES, FUNCTION, CATEGORY = ("GS_PLY",), ("switched_gs_ply",), "switch_axis_and_scale", "Comfy3D/Preprocessor"

    def switch_axis_and_scale(self, gs_ply, axis_x_to, axis_y_to, axis_z_to):
        order = {"x": 0, "y": 1, "z": 2}
        picks = [axis_x_to, axis_y_to, axis_z_to]
        if sorted(p[1] for p in picks) != ["x", "y", "z"]:
            _warn(self, "axis_x_to: %s, axis_y_to: %s, axis_z_to: %s have to be on separated axis" % tuple(picks))
            return (None,)
        target_axis, target_scale, inverts = [0, 0, 0], [0, 0, 0], 0
        for src, p in enumerate(picks):
            dst = order[p[1]]
            target_axis[dst] = src
            target_scale[src] = -1.0 if p[0] == "-" else 1.0
            inverts += p[0] == "-"
        # an odd permutation also inverts handedness
        perm_sign = 1 if target_axis in ([0, 1, 2], [1, 2, 0], [2, 0, 1]) else -1
        return (switch_ply_axis_and_scale(gs_ply, target_axis, target_scale, inverts + (perm_sign < 0)),)


class Stack_Orbit_Camera_Poses:
    """Cartesian product of six (start, stop, step) ranges -> list of [radius, elevation, azimuth, cx, cy, cz].
    Azimuth is circular: with start > stop and a positive step it runs up to +180, wraps to -180 and continues to `stop`."""
    @classmethod
    def INPUT_TYPES(cls):
        e = dict(min=ELEVATION_MIN, max=ELEVATION_MAX, step=0.0001)
        z = dict(min=AZIMUTH_MIN, max=AZIMUTH_MAX, step=0.0001)
        s = dict(step=0.0001)
        return {"required": {
            "orbit_radius_start": _F(1.75, **s), "orbit_radius_stop": _F(1.75, **s), "orbit_radius_step": _F(0.1, **s),
            "elevation_start": _F(0.0, **e), "elevation_stop": _F(0.0, **e), "elevation_step": _F(0.0, **e),
            "azimuth_start": _F(0.0, **z), "azimuth_stop": _F(0.0, **z), "azimuth_step": _F(0.0, **z),
            "orbit_center_X_start": _F(0.0, **s), "orbit_center_X_stop": _F(0.0, **s), "orbit_center_X_step": _F(0.1, **s),
            "orbit_center_Y_start": _F(0.0, **s), "orbit_center_Y_stop": _F(0.0, **s), "orbit_center_Y_step": _F(0.1, **s),
            "orbit_center_Z_start": _F(0.0, **s), "orbit_center_Z_stop": _F(0.0, **s), "orbit_center_Z_step": _F(0.1, **s)}}
    RETURN_TYPES = ("ORBIT_CAMPOSES", "FLOAT", "FLOAT", "FLOAT", "FLOAT", "FLOAT", "FLOAT")
    RETURN_NAMES = ("orbit_camposes", "orbit_radius_list", "elevation_list", "azimuth_list", "orbit_center_X_list", "orbit_center_Y_list", "orbit_center_Z_list")
    OUTPUT_IS_LIST = (False, True, True, True, True, True, True)
    FUNCTION, CATEGORY = "get_camposes", "Comfy3D/Preprocessor"

    @staticmethod
    def _values(start, stop, step, lo=-math.inf, hi=math.inf, linear=True):
        if abs(step) < 0.0001:
            step = -0.0001 if step < 0 else 0.0001
        if linear and ((step > 0 and stop < start) or (step < 0 and stop > start)):
            step = -step
        out, p, wrapped = [], start, False
        direct = linear or (step > 0 and start < stop) or (step < 0 and start > stop)
        while True:
            if direct:
                if (step > 0 and p > stop) or (step < 0 and p < stop):
                    break
            else:   # circular range that has to cross the min/max seam before it can reach `stop`
                if step > 0 and p > hi:
                    p, wrapped = lo + p % hi, True
                elif step < 0 and p < lo:
                    p, wrapped = hi + p % lo, True
                if wrapped and ((step > 0 and p > stop) or (step < 0 and p < stop)):
                    break
            out.append(p)
            p += step
        return out

    def get_camposes(self, orbit_radius_start, orbit_radius_stop, orbit_radius_step, elevation_start, elevation_stop, elevation_step,
                     azimuth_start, azimuth_stop, azimuth_step, orbit_center_X_start, orbit_center_X_stop, orbit_center_X_step,
                     orbit_center_Y_start, orbit_center_Y_stop, orbit_center_Y_step, orbit_center_Z_start, orbit_center_Z_stop, orbit_center_Z_step):
        axes = [self._values(orbit_radius_start, orbit_radius_stop, orbit_radius_step),
                self._values(elevation_start, elevation_stop, elevation_step, ELEVATION_MIN, ELEVATION_MAX),
                self._values(azimuth_start, azimuth_stop, azimuth_step, AZIMUTH_MIN, AZIMUTH_MAX, linear=False),
                self._values(orbit_center_X_start, orbit_center_X_stop, orbit_center_X_step),
                self._values(orbit_center_Y_start, orbit_center_Y_stop, orbit_center_Y_step),
                self._values(orbit_center_Z_start, orbit_center_Z_stop, orbit_center_Z_step)]
        poses = [[]]
        for vals in reversed(axes):   # radius varies slowest, orbit centre Z fastest (the reference's recursion order)
            poses = [[v] + p for v in vals for p in poses]
        cols = [[p[i] for p in poses] for i in range(6)]
        return (poses, *cols)


class Mesh_Orbit_Renderer:
    @classmethod
    def INPUT_TYPES(cls):
        c = dict(min=0.0, max=1.0, step=0.001)
        return {"required": {"mesh": ("MESH",), "render_image_width": _I(1024, min=128, max=8192), "render_image_height": _I(1024, min=128, max=8192),
                             "render_orbit_camera_poses": _POSES, "render_orbit_camera_fovy": _F(49.1, min=0.0, max=180.0, step=0.1),
                             "render_background_color_r": _F(0.0, **c), "render_background_color_g": _F(0.0, **c), "render_background_color_b": _F(0.0, **c),
                             "force_cuda_rasterize": ("BOOLEAN", {"default": True})},
                "optional": {"render_depth": ("BOOLEAN", {"default": False}), "render_normal": ("BOOLEAN", {"default": False})}}
    RETURN_TYPES = ("IMAGE", "MASK", "IMAGE", "IMAGE", "IMAGE")
    RETURN_NAMES = ("rendered_mesh_images", "rendered_mesh_masks", "all_rendered_depths", "all_rendered_normals", "all_rendered_viewcos")
    FUNCTION, CATEGORY = "render_mesh", "Comfy3D/Preprocessor"

    def render_mesh(self, mesh, render_image_width, render_image_height, render_orbit_camera_poses, render_orbit_camera_fovy,
                    render_background_color_r, render_background_color_g, render_background_color_b, force_cuda_rasterize,
                    render_depth=False, render_normal=False):
        renderer = DiffRastRenderer(mesh, force_cuda_rasterize)
        kinds = (['depth'] if render_depth else []) + (['normal'] if render_normal else [])
        ctl = DiffMeshCameraController(renderer, render_image_width, render_image_height, render_orbit_camera_fovy,
                                       static_bg=[render_background_color_r, render_background_color_g, render_background_color_b])
        images, masks, extra = ctl.render_all_pose(render_orbit_camera_poses, optional_render_types=kinds)
        depths = extra['depth'].repeat(1, 1, 1, 3) if 'depth' in extra else None           # [N,H,W,1] -> [N,H,W,3]
        return (images, masks.squeeze(-1), depths, extra.get('normal'), extra.get('viewcos'))


class Gaussian_Splatting_Orbit_Renderer:
    @classmethod
    def INPUT_TYPES(cls):
        c = dict(min=0.0, max=1.0, step=0.001)
        return {"required": {"gs_ply": ("GS_PLY",), "render_image_width": _I(1024, min=128, max=8192), "render_image_height": _I(1024, min=128, max=8192),
                             "render_orbit_camera_poses": _POSES, "render_orbit_camera_fovy": _F(49.1, min=0.0, max=180.0, step=0.1),
                             "render_background_color_r": _F(0.0, **c), "render_background_color_g": _F(0.0, **c), "render_background_color_b": _F(0.0, **c)}}
    RETURN_TYPES = ("IMAGE", "MASK", "IMAGE")
    RETURN_NAMES = ("rendered_gs_images", "rendered_gs_masks", "rendered_gs_depths")
    FUNCTION, CATEGORY = "render_gs", "Comfy3D/Preprocessor"

    def render_gs(self, gs_ply, render_image_width, render_image_height, render_orbit_camera_poses, render_orbit_camera_fovy,
                  render_background_color_r, render_background_color_g, render_background_color_b):
        sh_degree, _ = calculate_max_sh_degree_from_gs_ply(gs_ply)
        renderer = GaussianSplattingRenderer(sh_degree=sh_degree)
        renderer.initialize(gs_ply)
        ctl = GaussianSplattingCameraController(renderer, render_image_width, render_image_height, render_orbit_camera_fovy,
                                                static_bg=[render_background_color_r, render_background_color_g, render_background_color_b])
        with torch.no_grad():        # nodes run without autograd (ComfyUI executes them in inference mode): the whole orbit is one batched call
            images, masks, extra = ctl.render_all_pose(render_orbit_camera_poses)
        depths = extra['depth'].permute(0, 2, 3, 1).repeat(1, 1, 1, 3) if 'depth' in extra else None
        return (images.permute(0, 2, 3, 1), masks.squeeze(1), depths)                    # [N,H,W,3], [N,H,W], [N,H,W,3]


def _counts_ok(node, images, masks, poses):
    if len(images) != len(masks):
        _warn(node, "Number of reference images %d does not equal to number of masks %d" % (len(images), len(masks)))
        return False
    if len(images) != len(poses):
        _warn(node, "Number of reference images %d does not equal to number of reference camera poses %d" % (len(images), len(poses)))
        return False
    return True


class Gaussian_Splatting_3D:
    @classmethod
    def INPUT_TYPES(cls):
        lr = dict(min=0.000001, step=0.000001)
        return {"required": {
            "reference_images": ("IMAGE",), "reference_masks": ("MASK",), "reference_orbit_camera_poses": _POSES,
            "reference_orbit_camera_fovy": _F(49.1, min=0.0, max=180.0, step=0.1),
            "training_iterations": _I(30_000, min=1, max=_BIG), "batch_size": _I(1, min=1, max=_BIG),
            "ms_ssim_loss_weight": _F(0.2, min=0.0, max=1.0), "alpha_loss_weight": _F(3, min=0.0), "offset_loss_weight": _F(0.0, min=0.0),
            "offset_opacity_loss_weight": _F(0.0, min=0.0), "invert_background_probability": _F(0.5, min=0.0, max=1.0, step=0.1),
            "feature_learning_rate": _F(0.0025, **lr), "opacity_learning_rate": _F(0.05, **lr), "scaling_learning_rate": _F(0.005, **lr),
            "rotation_learning_rate": _F(0.001, **lr), "position_learning_rate_init": _F(0.00016, **lr),
            "position_learning_rate_final": _F(0.0000016, min=0.0000001, step=0.0000001), "position_learning_rate_delay_mult": _F(0.01, **lr),
            "position_learning_rate_max_steps": _I(30_000, min=1, max=_BIG), "initial_gaussians_num": _I(10_000, min=1, max=_BIG),
            "K_nearest_neighbors": _I(3, min=1, max=_BIG), "percent_dense": _F(0.01, min=0.00001, step=0.00001),
            "density_start_iterations": _I(500, min=0, max=_BIG), "density_end_iterations": _I(15_000, min=0, max=_BIG),
            "densification_interval": _I(100, min=1, max=_BIG), "opacity_reset_interval": _I(3000, min=1, max=_BIG),
            "densify_grad_threshold": _F(0.0002, min=0.00001, step=0.00001), "gaussian_sh_degree": _I(3, min=0)},
            "optional": {"points_cloud_to_initialize_gaussian": ("POINTCLOUD",), "ply_to_initialize_gaussian": ("GS_PLY",),
                         "mesh_to_initialize_gaussian": ("MESH",)}}
    RETURN_TYPES, RETURN_NAMES, FUNCTION, CATEGORY = ("GS_PLY",), ("gs_ply",), "run_gs", "Comfy3D/Algorithm"

    def run_gs(self, reference_images, reference_masks, reference_orbit_camera_poses, reference_orbit_camera_fovy, training_iterations, batch_size,
               ms_ssim_loss_weight, alpha_loss_weight, offset_loss_weight, offset_opacity_loss_weight, invert_background_probability,
               feature_learning_rate, opacity_learning_rate, scaling_learning_rate, rotation_learning_rate, position_learning_rate_init,
               position_learning_rate_final, position_learning_rate_delay_mult, position_learning_rate_max_steps, initial_gaussians_num,
               K_nearest_neighbors, percent_dense, density_start_iterations, density_end_iterations, densification_interval,
               opacity_reset_interval, densify_grad_threshold, gaussian_sh_degree, points_cloud_to_initialize_gaussian=None,
               ply_to_initialize_gaussian=None, mesh_to_initialize_gaussian=None):
        if not _counts_ok(self, reference_images, reference_masks, reference_orbit_camera_poses):
            return (None,)
        if batch_size > len(reference_images):
            _warn(self, "Batch size %d is bigger than number of reference images %d! Set batch size to %d instead" % (batch_size, len(reference_images), len(reference_images)))
            batch_size = len(reference_images)
        # initialiser precedence of the reference (nodes.py:1294-1299): point cloud, else ply, else mesh (None -> random ball)
        if points_cloud_to_initialize_gaussian is not None:
            gs_init_input = points_cloud_to_initialize_gaussian
        elif ply_to_initialize_gaussian is not None:
            gs_init_input = ply_to_initialize_gaussian
        else:
            gs_init_input = mesh_to_initialize_gaussian
        with torch.inference_mode(False):
            p = GSParams(training_iterations, batch_size, ms_ssim_loss_weight, alpha_loss_weight, offset_loss_weight, offset_opacity_loss_weight,
                         invert_background_probability, feature_learning_rate, opacity_learning_rate, scaling_learning_rate, rotation_learning_rate,
                         position_learning_rate_init, position_learning_rate_final, position_learning_rate_delay_mult, position_learning_rate_max_steps,
                         initial_gaussians_num, K_nearest_neighbors, percent_dense, density_start_iterations, density_end_iterations,
                         densification_interval, opacity_reset_interval, densify_grad_threshold, gaussian_sh_degree)
            gs = GaussianSplatting3D(p, gs_init_input)
            gs.prepare_training(reference_images, reference_masks, reference_orbit_camera_poses, reference_orbit_camera_fovy)
            gs.training()
            return (gs.renderer.gaussians.to_ply(),)


class Fitting_Mesh_With_Multiview_Images:
    def __init__(self):
        self.need_update = False

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "reference_images": ("IMAGE",), "reference_masks": ("MASK",), "reference_orbit_camera_poses": _POSES,
            "reference_orbit_camera_fovy": _F(49.1, min=0.0, max=180.0, step=0.1), "mesh": ("MESH",),
            "mesh_albedo_width": _I(1024, min=128, max=8192), "mesh_albedo_height": _I(1024, min=128, max=8192),
            "training_iterations": _I(1024, min=1, max=100000), "batch_size": _I(3, min=1, max=_BIG),
            "texture_learning_rate": _F(0.001, min=0.00001, step=0.00001), "train_mesh_geometry": ("BOOLEAN", {"default": False}),
            "geometry_learning_rate": _F(0.0001, min=0.00001, step=0.00001), "ms_ssim_loss_weight": _F(0.5, min=0.0, max=1.0, step=0.01),
            "remesh_after_n_iteration": _I(512, min=128, max=100000), "invert_background_probability": _F(0.5, min=0.0, max=1.0, step=0.1),
            "force_cuda_rasterize": ("BOOLEAN", {"default": True})}}
    RETURN_TYPES, RETURN_NAMES, FUNCTION, CATEGORY = ("MESH", "IMAGE"), ("trained_mesh", "baked_texture"), "fitting_mesh", "Comfy3D/Algorithm"

    def fitting_mesh(self, reference_images, reference_masks, reference_orbit_camera_poses, reference_orbit_camera_fovy, mesh, mesh_albedo_width,
                     mesh_albedo_height, training_iterations, batch_size, texture_learning_rate, train_mesh_geometry, geometry_learning_rate,
                     ms_ssim_loss_weight, remesh_after_n_iteration, invert_background_probability, force_cuda_rasterize):
        if mesh.vt is None:
            raise NotImplementedError("mesh.auto_uv() (xatlas, CPU asset tooling) is out of scope: provide a mesh with UVs")
        mesh.set_new_albedo(mesh_albedo_width, mesh_albedo_height)
        if not _counts_ok(self, reference_images, reference_masks, reference_orbit_camera_poses):
            return (None, None)
        if batch_size > len(reference_images):
            _warn(self, "Batch size %d is bigger than number of reference images %d! Set batch size to %d instead" % (batch_size, len(reference_images), len(reference_images)))
            batch_size = len(reference_images)
        with torch.inference_mode(False):
            fitter = DiffMesh(mesh, training_iterations, batch_size, texture_learning_rate, train_mesh_geometry, geometry_learning_rate,
                              ms_ssim_loss_weight, remesh_after_n_iteration, invert_background_probability, force_cuda_rasterize)
            fitter.prepare_training(reference_images, reference_masks, reference_orbit_camera_poses, reference_orbit_camera_fovy)
            fitter.training()
            out = fitter.get_mesh_and_texture()
        if fitter.remesh_skipped:      # the reference remeshes every remesh_after_n_iteration steps (pymeshlab, diff_mesh.py:132-141): not part of this implementation
            return {"ui": {"text": ["Fitting_Mesh_With_Multiview_Images: the periodic remesh (every %d iterations) was SKIPPED -- the returned mesh keeps the input "
                                    "topology (%d vertices, %d faces)" % (remesh_after_n_iteration, int(out[0].v.shape[0]), int(out[0].f.shape[0]))]}, "result": out}
        return out


NODE_CLASS_MAPPINGS = {"[Comfy3D] " + n.replace("_", " "): c for n, c in
                       [(c.__name__, c) for c in (Load_3DGS, Save_3DGS, Switch_3DGS_Axis, Stack_Orbit_Camera_Poses, Mesh_Orbit_Renderer,
                                                  Gaussian_Splatting_Orbit_Renderer, Gaussian_Splatting_3D, Fitting_Mesh_With_Multiview_Images)]}
NODE_DISPLAY_NAME_MAPPINGS = {k: k.replace("[Comfy3D] ", "") for k in NODE_CLASS_MAPPINGS}

"""Orbit cameras and the per-view controller used by the 3DGS / mesh renderers.

Host-side mirror of the reference's camera layer (same public names, argument meaning and
conventions) -- /root/reference/shared_utils/camera_utils.py:
  ORBITPOSE_PRESET_DICT :12-21, OrbitCamera :89-165, get_projection_matrix :174-185,
  MiniCam :188-214, BaseCameraController :216-274, compose_orbit_camposes :276-287.
`orbit_camera` restates kiui.cam.orbit_camera (kiui 0.2.14, not vendored by the reference;
my-reqs.txt:50): x = r cos(e) sin(a), y = -r sin(e), z = r cos(e) cos(a), OpenGL look-at.
Own implementation; nothing here touches the GPU except the three small per-view uploads.
"""
from abc import ABC, abstractmethod
from collections import OrderedDict
import math

import numpy as np
import torch

ORBITPOSE_PRESET_DICT = OrderedDict([
    ("Custom",          [[0.0, 90.0, 0.0, 0.0, -90.0, 0.0], [-90.0, 0.0, 180.0, 90.0, 0.0, 0.0]]),
    ("CRM(6)",          [[0.0, 90.0, 0.0, 0.0, -90.0, 0.0], [-90.0, 0.0, 180.0, 90.0, 0.0, 0.0]]),
    ("Wonder3D(6)",     [[0.0] * 6, [0.0, 45.0, 90.0, 180.0, -90.0, -45.0]]),
    ("Zero123Plus(6)",  [[-20.0, 10.0, -20.0, 10.0, -20.0, 10.0], [30.0, 90.0, 150.0, -150.0, -90.0, -30.0]]),
    ("Era3D(6)",        [[0.0] * 6, [0.0, 45.0, 90.0, 180.0, -90.0, -45.0]]),
    ("MVDream(4)",      [[0.0] * 4, [0.0, 90.0, 180.0, -90.0]]),
    ("Unique3D(4)",     [[0.0] * 4, [0.0, 90.0, 180.0, -90.0]]),
    ("CharacterGen(4)", [[0.0] * 4, [-90.0, 180.0, 90.0, 0.0]]),
])
ELEVATION_MIN, ELEVATION_MAX = -89.999, 89.999
AZIMUTH_MIN, AZIMUTH_MAX = -180.0, 180.0


def _unit(v, eps=1e-20):
    return v / np.sqrt(np.maximum(np.sum(v * v, axis=-1, keepdims=True), eps))


def look_at(campos, target, opengl=True):
    """3x3 rotation whose columns are (right, up, forward); OpenGL: the camera looks down -z."""
    up0 = np.array([0, 1, 0], dtype=np.float32)
    if opengl:
        fwd = _unit(campos - target)
        right = _unit(np.cross(up0, fwd))
        up = _unit(np.cross(fwd, right))
    else:
        fwd = _unit(target - campos)
        right = _unit(np.cross(fwd, up0))
        up = _unit(np.cross(right, fwd))
    return np.stack([right, up, fwd], axis=1)


def orbit_camera(elevation, azimuth, radius=1, is_degree=True, target=None, opengl=True):
    """camera-to-world 4x4 (float32) of a camera orbiting `target`; elevation < 0 looks from above."""
    if is_degree:
        elevation, azimuth = np.deg2rad(elevation), np.deg2rad(azimuth)
    if target is None:
        target = np.zeros([3], dtype=np.float32)
    pos = np.array([radius * np.cos(elevation) * np.sin(azimuth), -radius * np.sin(elevation),
                    radius * np.cos(elevation) * np.cos(azimuth)]) + target
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = look_at(pos, target, opengl)
    c2w[:3, 3] = pos
    return c2w


def calculate_fovX(H, W, fovy):
    return 2 * np.arctan(np.tan(fovy / 2) * W / H)


def _axis_angle(rotvec):
    """rotation matrix of a rotation vector (Rodrigues), float64"""
    rotvec = np.asarray(rotvec, dtype=np.float64)
    a = float(np.linalg.norm(rotvec))
    if a < 1e-300:
        return np.eye(3)
    k = rotvec / a
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


class OrbitCamera:
    """Size, fovy/fovx (radians), near/far and perspective are what the renderers read; pose / view / mvp and the orbit / scale / pan
    controls follow the reference's interactive camera (camera_utils.py:105-169), with the orientation kept as a plain 3x3 matrix."""

    def __init__(self, W, H, r=2, fovy=60, near=0.01, far=100):
        self.W, self.H, self.radius = W, H, r
        self.fovy = np.deg2rad(fovy)
        self.near, self.far = near, far
        self.center = np.zeros(3, dtype=np.float32)
        self.rot_matrix = np.eye(3)
        self.up = np.array([0, 1, 0], dtype=np.float32)

    @property
    def fovx(self):
        return calculate_fovX(self.H, self.W, self.fovy)

    @property
    def pose(self):
        """camera-to-world: back off by the radius along the camera's z, rotate, then shift by -center (as the reference does)"""
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, :3] = self.rot_matrix
        c2w[:3, 3] = self.rot_matrix[:, 2] * self.radius - self.center
        return c2w

    @property
    def campos(self):
        return self.pose[:3, 3]

    @property
    def view(self):
        return np.linalg.inv(self.pose)

    @property
    def mvp(self):
        return self.perspective @ np.linalg.inv(self.pose)

    def orbit(self, dx, dy):
        """mouse drag: 0.05 degrees per unit about the world up axis (dx) and the camera's side axis (dy)"""
        side = self.rot_matrix[:, 0]
        self.rot_matrix = _axis_angle(self.up * np.radians(-0.05 * dx)) @ _axis_angle(side * np.radians(-0.05 * dy)) @ self.rot_matrix

    def scale(self, delta):
        self.radius *= 1.1 ** (-delta)

    def pan(self, dx, dy, dz=0):
        self.center += (0.0005 * self.rot_matrix @ np.array([-dx, -dy, dz])).astype(np.float32)

    @property
    def perspective(self):
        """OpenGL projection with y flipped (image row 0 = top), as the mesh renderer expects."""
        t = np.tan(self.fovy / 2)
        n, f = self.near, self.far
        P = np.zeros((4, 4), dtype=np.float32)
        P[0, 0] = 1 / (t * self.W / self.H)
        P[1, 1] = -1 / t
        P[2, 2] = -(f + n) / (f - n)
        P[2, 3] = -(2 * f * n) / (f - n)
        P[3, 2] = -1
        return P

    @property
    def intrinsics(self):
        focal = self.H / (2 * np.tan(self.fovy / 2))
        return np.array([focal, focal, self.W // 2, self.H // 2], dtype=np.float32)


def get_projection_matrix(znear, zfar, fovX, fovY, z_sign=1.0):
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / math.tan(fovX / 2)
    P[1, 1] = 1 / math.tan(fovY / 2)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


class MiniCam:
    """COLMAP-style matrices for the 3DGS rasterizer from an OpenGL c2w pose.

    Keeps the reference's conventions exactly (camera_utils.py:197-214): rows 1-2 of the rotation
    and the whole translation of w2c are negated, matrices are stored transposed (row-vector
    convention) and camera_center = -c2w[:3,3].
    """

    def __init__(self, c2w, width, height, fovy, fovx, znear, zfar, projection_matrix=None, device="cuda"):
        self.image_width, self.image_height = width, height
        self.FoVy, self.FoVx, self.znear, self.zfar = fovy, fovx, znear, zfar
        w2c = np.linalg.inv(c2w)
        w2c[1:3, :3] *= -1
        w2c[:3, 3] *= -1
        self.world_view_transform = torch.tensor(w2c).transpose(0, 1).to(device)
        if projection_matrix is None:
            projection_matrix = get_projection_matrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).transpose(0, 1).to(device)
        self.projection_matrix = projection_matrix
        self.full_proj_transform = self.world_view_transform @ self.projection_matrix
        self.camera_center = -torch.tensor(c2w[:3, 3]).to(device)


class BaseCameraController(ABC):
    def __init__(self, renderer, cam_size_W, cam_size_H, reference_orbit_camera_fovy, invert_bg_prob=1.0, static_bg=None, device='cuda'):
        self.device = torch.device(device)
        self.renderer = renderer
        self.cam = OrbitCamera(cam_size_W, cam_size_H, fovy=reference_orbit_camera_fovy)
        self.invert_bg_prob = invert_bg_prob
        self.black_bg = torch.zeros(3, dtype=torch.float32, device=self.device)
        self.white_bg = torch.ones(3, dtype=torch.float32, device=self.device)
        self.static_bg = None if static_bg is None else torch.tensor(static_bg, dtype=torch.float32, device=self.device)
        self.post_init()
        super().__init__()

    def post_init(self):
        pass

    @abstractmethod
    def get_render_result(self, render_pose, bg_color, **kwargs):
        ...

    def render_at_pose(self, cam_pose, **kwargs):
        radius, elevation, azimuth, cx, cy, cz = cam_pose
        pose = orbit_camera(elevation, azimuth, radius, target=np.array([cx, cy, cz], dtype=np.float32))
        if self.static_bg is not None:
            bg = self.static_bg
        else:
            bg = self.white_bg if np.random.rand() > self.invert_bg_prob else self.black_bg
        return self.get_render_result(pose, bg, **kwargs)

    def render_all_pose(self, all_cam_poses, **kwargs):
        """-> images [V,3,H,W], masks [V,1,H,W], dict of every other per-view output stacked on dim 0"""
        images, masks, extra = [], [], {}
        for cam_pose in all_cam_poses:
            out = self.render_at_pose(cam_pose, **kwargs)
            images.append(out["image"])
            masks.append(out["alpha"])
            for k, v in out.items():
                extra.setdefault(k, []).append(v)
        extra = {k: torch.stack(v, dim=0) for k, v in extra.items()}
        return torch.stack(images, dim=0), torch.stack(masks, dim=0), extra


def compose_orbit_camposes(orbit_radius, orbit_elevations, orbit_azimuths, orbit_center_x, orbit_center_y, orbit_center_z):
    return [[orbit_radius[i], float(np.clip(orbit_elevations[i], ELEVATION_MIN, ELEVATION_MAX)),
             float(np.clip(orbit_azimuths[i], AZIMUTH_MIN, AZIMUTH_MAX)),
             orbit_center_x[i], orbit_center_y[i], orbit_center_z[i]] for i in range(len(orbit_radius))]

"""ctypes signatures of include/c3d_mesh.h"""
import ctypes as C

vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t


class MeshView(C.Structure):
    """struct c3d_mesh_view"""
    _fields_ = [("V", i32), ("T", i32), ("Vt", i32), ("H", i32), ("W", i32), ("Ht", i32), ("Wt", i32), ("clip_from_world", C.c_float * 16), ("bg", C.c_float * 3)]


class MeshStepLoss(C.Structure):
    """struct c3d_mesh_step_loss"""
    _fields_ = [("w_mse", C.c_float), ("w_ssim", C.c_float), ("scale", C.c_float)]


SIGNATURES = {
    "c3d_mesh_raster_scratch_bytes": (sz, [i32, i32, i32, i32]),
    "c3d_mesh_rasterize_fwd": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "c3d_mesh_rasterize_peel_fwd": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "c3d_mesh_rasterize_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "c3d_mesh_vertex_topology_bytes": (sz, [i32, i32]),
    "c3d_mesh_build_vertex_topology": (C.c_int, [vp, i32, i32, vp, vp]),
    "c3d_mesh_rasterize_bwd_scratch_bytes": (sz, [i32, i32]),
    "c3d_mesh_rasterize_bwd_gather": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "c3d_mesh_interpolate_fwd": (C.c_int, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "c3d_mesh_interpolate_bwd": (C.c_int, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "c3d_mesh_interpolate_da_bwd": (C.c_int, [vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
    "c3d_mesh_texture_fwd": (C.c_int, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "c3d_mesh_texture_bwd": (C.c_int, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "c3d_mesh_mip_info": (i32, [i32, i32, i32, vp, vp]),
    "c3d_mesh_mip_build": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "c3d_mesh_mip_build_bwd": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "c3d_mesh_texture_mip_fwd": (C.c_int, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "c3d_mesh_texture_mip_bwd": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "c3d_mesh_transform_fwd": (C.c_int, [vp, vp, i32, vp, vp]),
    "c3d_mesh_transform_bwd": (C.c_int, [vp, vp, i32, vp, vp]),
    "c3d_mesh_shade_fwd": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp]),
    "c3d_mesh_shade_bwd": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]),
    "c3d_mesh_antialias_scratch_bytes": (sz, [i32]),
    "c3d_mesh_antialias_build_topology": (C.c_int, [vp, i32, vp, vp]),
    "c3d_mesh_antialias_fwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "c3d_mesh_view_state_bytes": (sz, [i32, i32, i32, i32]),
    "c3d_mesh_view_bwd_scratch_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "c3d_mesh_view_fwd": (C.c_int, [C.POINTER(MeshView), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "c3d_mesh_view_bwd": (C.c_int, [C.POINTER(MeshView), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "c3d_mesh_step_workspace_bytes": (sz, [i32] * 8),
    "c3d_mesh_train_views": (C.c_int, [C.POINTER(MeshView), i32] + [vp] * 8 + [C.POINTER(vp), C.POINTER(vp), C.POINTER(MeshStepLoss), vp, vp, vp, i32, i32, vp, vp]),
    "c3d_mesh_antialias_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
}

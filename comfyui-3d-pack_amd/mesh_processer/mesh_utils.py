"""3DGS PLY wire format: the on-disk / inter-node format either side of the renderer.

Mirror of the schema functions of /root/reference/mesh_processer/mesh_utils.py:
  construct_list_of_gs_attributes :333-344   calculate_max_sh_degree_from_gs_ply :346-350   write_gs_ply :352-359
  read_gs_ply :361-390   switch_vector_axis / switch_ply_axis_and_scale :446-472
Properties, in order: x y z nx ny nz f_dc_0..2 f_rest_0..3(K-1)-1 opacity scale_0..2 rot_0..3, all float32; f_dc / f_rest are
stored CHANNEL-MAJOR ([3, K-1] flattened).  The PLY container comes from c3d_hip.ply (plyfile-compatible subset)."""
import numpy as np

from c3d_hip.ply import PlyData, PlyElement


def construct_list_of_gs_attributes(features_dc, features_rest, scaling, rotation):
    names = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    names += ['f_dc_%d' % i for i in range(features_dc.shape[1] * features_dc.shape[2])]
    names += ['f_rest_%d' % i for i in range(features_rest.shape[1] * features_rest.shape[2])]
    names.append('opacity')
    names += ['scale_%d' % i for i in range(scaling.shape[1])]
    names += ['rot_%d' % i for i in range(rotation.shape[1])]
    return names


def calculate_max_sh_degree_from_gs_ply(plydata):
    extra = [p.name for p in plydata.elements[0].properties if p.name.startswith("f_rest_")]
    return int(((len(extra) + 3) / 3) ** 0.5 - 1), extra


def write_gs_ply(xyz, normals, f_dc, f_rest, opacities, scale, rotation, list_of_attributes):
    cols = np.concatenate((xyz, normals, f_dc, f_rest, opacities, scale, rotation), axis=1).astype(np.float32)
    assert cols.shape[1] == len(list_of_attributes)
    el = np.empty(cols.shape[0], dtype=[(a, 'f4') for a in list_of_attributes])
    for j, a in enumerate(list_of_attributes):
        el[a] = cols[:, j]
    return PlyData([PlyElement.describe(el, 'vertex')])


def read_gs_ply(plydata):
    """-> xyz [N,3], features_dc [N,3,1], features_extra [N,3,K-1], opacities [N,1], scales [N,3], rots [N,4]  (float64, as the reference)"""
    v = plydata.elements[0]
    col = lambda n: np.asarray(v[n], dtype=np.float64)
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opacities = col("opacity")[..., np.newaxis]
    features_dc = np.stack((col("f_dc_0"), col("f_dc_1"), col("f_dc_2")), axis=1)[..., np.newaxis]
    deg, extra = calculate_max_sh_degree_from_gs_ply(plydata)
    features_extra = np.stack([col(n) for n in extra], axis=1) if extra else np.zeros((xyz.shape[0], 0))
    features_extra = features_extra.reshape((xyz.shape[0], 3, (deg + 1) ** 2 - 1))
    scales = np.stack([col(p.name) for p in v.properties if p.name.startswith("scale_")], axis=1)
    rots = np.stack([col(p.name) for p in v.properties if p.name.startswith("rot")], axis=1)
    return xyz, features_dc, features_extra, opacities, scales, rots


def switch_vector_axis(vector3s, target_axis):
    """vector3s[:, [0,1,2]] = vector3s[:, target_axis]"""
    return vector3s[:, list(target_axis)]


def _quat_to_axis_angle(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, v = q[:, :1], q[:, 1:]
    n = np.linalg.norm(v, axis=1, keepdims=True)
    half = np.arctan2(n, w)
    k = np.where(n > 1e-12, 2 * half / np.maximum(n, 1e-12), 2.0)
    return v * k


def _axis_angle_to_quat(a):
    ang = np.linalg.norm(a, axis=1, keepdims=True)
    k = np.where(ang > 1e-12, np.sin(ang / 2) / np.maximum(ang, 1e-12), 0.5)
    return np.concatenate([np.cos(ang / 2), a * k], axis=1)


def switch_ply_axis_and_scale(plydata, target_axis, target_scale, coordinate_invert_count):
    """permute / scale the coordinate axes of a 3DGS PLY (positions, scales and rotations); an odd number of axis inversions
    flips the rotation sense (reference :446-472)."""
    xyz, f_dc, f_extra, opac, scales, rots = read_gs_ply(plydata)
    ts = np.asarray(target_scale, dtype=np.float64)
    xyz = switch_vector_axis(xyz * ts, target_axis)
    scales = switch_vector_axis(scales, target_axis)
    aa = switch_vector_axis(_quat_to_axis_angle(rots) * ts, target_axis)
    if coordinate_invert_count % 2 != 0:
        aa = -aa
    rots = _axis_angle_to_quat(aa)
    return write_gs_ply(xyz, np.zeros_like(xyz), f_dc.reshape(f_dc.shape[0], -1), f_extra.reshape(f_extra.shape[0], -1), opac, scales, rots,
                        construct_list_of_gs_attributes(f_dc, f_extra, scales, rots))

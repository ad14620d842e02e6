"""Seeded synthetic workloads named in BASELINE.json / BASELINE.md section 3 (no assets ship with the
reference: SURVEY.md 0.2-iii).  numpy only; used by bench.py, the tests and the golden-vector script."""
import math

import numpy as np

from shared_utils.camera_utils import MiniCam, OrbitCamera, orbit_camera


def camera_settings(W, H, fovy_deg, elevation, azimuth, radius, bg=(1.0, 1.0, 1.0), sh_degree=3, target=(0, 0, 0),
                    near=0.01, far=100, scale_modifier=1.0):
    """Raster settings (numpy, host) for one orbit pose, built exactly the way the reference's controller
    builds them (camera_utils.py:240-251 -> main_3DGS.py:76-82 -> MiniCam)."""
    cam = OrbitCamera(W, H, fovy=fovy_deg, near=near, far=far)
    c2w = orbit_camera(elevation, azimuth, radius, target=np.array(target, dtype=np.float32))
    mc = MiniCam(c2w, W, H, cam.fovy, cam.fovx, near, far, device="cpu")
    return {
        "image_height": H, "image_width": W,
        "tanfovx": math.tan(mc.FoVx * 0.5), "tanfovy": math.tan(mc.FoVy * 0.5),
        "bg": np.asarray(bg, dtype=np.float32), "scale_modifier": scale_modifier,
        "viewmatrix": mc.world_view_transform.contiguous().numpy().astype(np.float32),
        "projmatrix": mc.full_proj_transform.contiguous().numpy().astype(np.float32),
        "sh_degree": sh_degree, "campos": mc.camera_center.numpy().astype(np.float32),
        "prefiltered": False, "debug": False,
    }


def orbit_poses_64():
    """configs 2-4: elevations {-30,0,30,60} x 16 azimuths, radius 2.2."""
    return [(2.2, float(e), float(a)) for e in (-30, 0, 30, 60) for a in np.arange(16) * 22.5]


def make_cloud(N, seed=1234, sh_degree=3, log_scale_mean=math.log(0.004), log_scale_std=0.5, radius=1.0,
               activated=True):
    """configs 2-4 cloud: xyz uniform in the unit ball, log-scale ~ N(log .004, .5), random unit quats,
    opacity_raw ~ N(0,1.5), f_dc ~ U(-1.5,1.5), f_rest ~ N(0,.1).  activated=True returns what the rasterizer
    consumes (exp / sigmoid / normalised)."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = (d * (radius * np.cbrt(rng.uniform(size=(N, 1))))).astype(np.float32)
    log_scale = rng.normal(log_scale_mean, log_scale_std, size=(N, 3)).astype(np.float32)
    rot = rng.normal(size=(N, 4)).astype(np.float32)
    opacity_raw = rng.normal(0, 1.5, size=(N, 1)).astype(np.float32)
    K = (sh_degree + 1) ** 2
    shs = np.empty((N, K, 3), dtype=np.float32)
    shs[:, 0] = rng.uniform(-1.5, 1.5, size=(N, 3))
    if K > 1:
        shs[:, 1:] = rng.normal(0, 0.1, size=(N, K - 1, 3))
    out = {"means3D": xyz, "shs": shs}
    if activated:
        out["scales"] = np.exp(log_scale)
        out["rotations"] = rot / np.linalg.norm(rot, axis=1, keepdims=True)
        out["opacities"] = (1.0 / (1.0 + np.exp(-opacity_raw))).astype(np.float32)
    else:
        out.update(scales=log_scale, rotations=rot, opacities=opacity_raw)
    return out


def make_ball_cloud(N=10000, seed=0, sh_degree=3, scale=0.02, opacity=0.1, radius=0.5):
    """config 1: the reference's random-ball init (main_3DGS_renderer.py:811-826) with isotropic scale 0.02
    standing in for distCUDA2, identity rotations, opacity 0.1, SH deg 3 with zero rest."""
    rng = np.random.default_rng(seed)
    phis = rng.uniform(size=N) * 2 * np.pi
    costheta = rng.uniform(size=N) * 2 - 1
    thetas = np.arccos(costheta)
    r = radius * np.cbrt(rng.uniform(size=N))
    xyz = np.stack([r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis), r * np.cos(thetas)], axis=1)
    K = (sh_degree + 1) ** 2
    shs = np.zeros((N, K, 3), dtype=np.float32)
    shs[:, 0] = rng.uniform(size=(N, 3)) / 255.0   # colours U/255 stored straight into the DC band, as the reference does
    rot = np.zeros((N, 4), dtype=np.float32); rot[:, 0] = 1
    return {"means3D": xyz.astype(np.float32), "shs": shs, "scales": np.full((N, 3), scale, dtype=np.float32),
            "rotations": rot, "opacities": np.full((N, 1), opacity, dtype=np.float32)}


def make_small_scene(N=48, seed=7, sh_degree=3, scale=0.08, spread=0.6):
    """tiny, well-conditioned scene for gradient checks (every Gaussian covers several pixels)."""
    rng = np.random.default_rng(seed)
    K = (sh_degree + 1) ** 2
    rot = rng.normal(size=(N, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    shs = np.concatenate([rng.uniform(-1.0, 1.5, size=(N, 1, 3)), rng.normal(0, 0.3, size=(N, K - 1, 3))], axis=1)
    return {"means3D": rng.uniform(-spread, spread, size=(N, 3)).astype(np.float32),
            "shs": shs.astype(np.float32),
            "scales": np.exp(rng.normal(math.log(scale), 0.4, size=(N, 3))).astype(np.float32),
            "rotations": rot.astype(np.float32),
            "opacities": rng.uniform(0.05, 0.95, size=(N, 1)).astype(np.float32)}


# ---------------------------------------------------------------------------------------------- mesh workloads
def make_uv_sphere(n_lat=24, n_lon=48, radius=0.6, displacement=0.05, seed=5):
    """lat-long sphere displaced by seeded low-frequency noise (BASELINE config 5 uses 500x500 -> 499,000 triangles).
    Returns v [V,3] f32, f [T,3] i32, vt [V,2] f32 (lat-long UVs, one per vertex), vn [V,3] f32."""
    rng = np.random.default_rng(seed)
    lat = np.linspace(0.0, np.pi, n_lat + 1)[1:-1]                      # interior rings
    lon = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)
    th, ph = np.meshgrid(lat, lon, indexing="ij")
    dirs = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    dirs = np.concatenate([dirs, [[0, 1, 0]], [[0, -1, 0]]], 0)
    k = rng.normal(size=(4, 3)) * 2.0
    phs = rng.uniform(0, 2 * np.pi, size=4)
    disp = sum(np.sin(dirs @ k[i] + phs[i]) for i in range(4)) / 4.0
    v = dirs * (radius * (1.0 + displacement * disp))[:, None]
    R = n_lat - 1
    idx = lambda r, c: r * n_lon + (c % n_lon)
    f = []
    for r in range(R - 1):
        for c in range(n_lon):
            f.append([idx(r, c), idx(r + 1, c), idx(r, c + 1)])
            f.append([idx(r, c + 1), idx(r + 1, c), idx(r + 1, c + 1)])
    top, bot = R * n_lon, R * n_lon + 1
    for c in range(n_lon):
        f.append([top, idx(0, c), idx(0, c + 1)])
        f.append([bot, idx(R - 1, c + 1), idx(R - 1, c)])
    vt = np.concatenate([np.stack([ph / (2 * np.pi), th / np.pi], -1).reshape(-1, 2), [[0.5, 0.0]], [[0.5, 1.0]]], 0)
    vn = v / np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32), np.asarray(f, dtype=np.int32), vt.astype(np.float32), vn.astype(np.float32)


def mesh_clip_positions(v, elevation, azimuth, radius, W, H, fovy_deg=49.1, near=0.01, far=100):
    """clip-space vertices exactly as DiffRastRenderer.render builds them (diff_mesh_renderer.py:90-95):
    v_cam = [v,1] @ inv(pose)^T ; v_clip = v_cam @ proj^T with proj = OrbitCamera.perspective (y flipped)."""
    cam = OrbitCamera(W, H, fovy=fovy_deg, near=near, far=far)
    pose = orbit_camera(elevation, azimuth, radius).astype(np.float32)
    vh = np.concatenate([v, np.ones((v.shape[0], 1), np.float32)], 1)
    v_cam = vh @ np.linalg.inv(pose).T.astype(np.float32)
    v_clip = v_cam @ cam.perspective.T
    return v_clip.astype(np.float32)[None], v_cam.astype(np.float32)[None], pose

"""TEST INFRASTRUCTURE: builds oracle/_ref/ from reference sources where they lie under /root/reference (nothing is copied).

The only compilable source in the reference that computes a function of this path is the CPU half of Hunyuan3D's
`custom_rasterizer` (Gen_3D_Modules/Hunyuan3D_V2/hy3dgen/texgen/custom_rasterizer/lib/custom_rasterizer_kernel/rasterizer.cpp:94-133),
the stand-alone `rasterize(pos, tri, resolution) -> (findices, barycentric)` its texture-baking renderer uses in place of
`dr.rasterize` (differentiable_renderer/mesh_render.py:165-176).  It is one translation unit + torch headers; its CUDA half and the
hierarchy builder are replaced by throwing stubs (oracle/hy_ref_stub.cpp), its CUDA context include by an empty header
(oracle/ref_shim/).  The reference's own setup.py is not run.

oracle/_ref/ is git-ignored and travels to the GPU box with the snapshot; tests that use it skip when it is absent, and the
comparison is also frozen as tests/golden/mesh_hy_raster.npz (tests/golden/make_golden_mesh.py).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/Gen_3D_Modules/Hunyuan3D_V2/hy3dgen/texgen/custom_rasterizer/lib/custom_rasterizer_kernel"
OUT_DIR = os.path.join(HERE, "_ref")
MODULE = "hy_custom_rasterizer_ref"
OUT = os.path.join(OUT_DIR, MODULE + ".so")


def available():
    return os.path.exists(OUT)


def build(force=False):
    """g++ on rasterizer.cpp + the stub; returns the .so path, or None when /root/reference is absent (GPU box) and nothing is prebuilt."""
    src = os.path.join(REF_SRC, "rasterizer.cpp")
    if not os.path.exists(src):
        return OUT if available() else None
    stub = os.path.join(HERE, "hy_ref_stub.cpp")
    if available() and not force and os.path.getmtime(OUT) >= max(os.path.getmtime(src), os.path.getmtime(stub)):
        return OUT
    import torch
    tdir = os.path.dirname(torch.__file__)
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-w", "-D__host__=", "-D__device__=",
           "-DTORCH_EXTENSION_NAME=" + MODULE, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(HERE, "ref_shim"), "-I" + REF_SRC,
           "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I" + sysconfig.get_paths()["include"],
           src, stub, "-o", OUT,
           "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
           "-Wl,-rpath," + os.path.join(tdir, "lib")]
    subprocess.run(cmd, check=True)
    return OUT


def load():
    """import the compiled reference module (torch must be imported first: the extension links libtorch)."""
    import importlib.util
    import torch  # noqa: F401
    spec = importlib.util.spec_from_file_location(MODULE, OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rasterize(pos, tri, resolution):
    """The reference wrapper's call (custom_rasterizer/render.py:19-23) on CPU tensors: pos [1,V,4] clip space, tri [T,3] int32,
    resolution (H, W) -> findices [H,W] int32 (face + 1, 0 = empty), barycentric [H,W,3]."""
    import torch
    m = load()
    findices, bary = m.rasterize_image(pos[0].contiguous(), tri.contiguous(), torch.zeros(0), int(resolution[1]), int(resolution[0]), 1e-6, 0)
    return findices, bary


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print(p if p else "no /root/reference and nothing prebuilt")

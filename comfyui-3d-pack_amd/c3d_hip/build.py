"""Build libc3d_hip.so (the C-ABI library of include/c3d_*.h) for gfx950 with hipcc, in-tree.

Usage: python build.py [--force].  hipcc cross-compiles without a GPU; the .so lands in
comfyui-3d-pack_amd/lib/ and travels with the tree (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(PKG, "build")
LIB = os.path.join(LIBDIR, "libc3d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc -O3 packs neighbouring scalar f32 multiplies / adds into v_pk_*_f32 (+ moves and s_nop padding); on gfx950 a packed op
# issues in 4.3-4.9 cycles against 2.9 for a plain one (profiles/r01f_valu_rate_microbench.txt), and the compositing kernels are issue bound:
# forward compositing 0.182 -> 0.164 ms, backward 0.497 -> 0.479 ms, per-Gaussian pass 0.88 -> 0.81 ms without it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize"] + os.environ.get("C3D_EXTRA_HIPCC_FLAGS", "").split() + [
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(PKG), "include")
    headers = sorted("../../include/" + f for f in os.listdir(inc) if f.endswith(".h")) if os.path.isdir(inc) else []   # every C-ABI header
    for f in sorted(os.listdir(CSRC)) + headers:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libc3d_hip.digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    srcs = _sources()

    def cc(f):
        obj = os.path.join(OBJDIR, f[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, f), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (f, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    open(stamp, "w").write(dig)
    return LIB


def code_digest():
    """sha256 over csrc/*, include/*.h and the compile flags: identifies the kernel code a measurement belongs to (the GPU box has no .git)"""
    return _digest()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

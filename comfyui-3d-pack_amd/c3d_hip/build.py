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


def _current(stamp, dig):
    try:
        return os.path.exists(LIB) and open(stamp).read() == dig
    except OSError:
        return False


def build(force=False, verbose=False):
    """Several processes may call this at once (every rank of a multi-GPU launch imports c3d_hip): the rebuild runs under an exclusive file lock, objects
    go to a per-process directory, and both the library and its digest stamp are written to temporary names and renamed into place -- a concurrent
    CDLL sees either the old file or the complete new one, never a half-written library."""
    import fcntl
    import shutil
    import tempfile
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libc3d_hip.digest")
    dig = _digest()
    if not force and _current(stamp, dig):
        return LIB
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _current(stamp, dig):      # another process built it while this one waited for the lock
                return LIB
            objdir = tempfile.mkdtemp(prefix="obj.", dir=OBJDIR)
            try:
                return _build_locked(objdir, stamp, dig, verbose)
            finally:
                shutil.rmtree(objdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(objdir, stamp, dig, verbose):
    srcs = _sources()

    def cc(f):
        obj = os.path.join(objdir, f[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, f), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (f, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    tmp_lib = os.path.join(objdir, "libc3d_hip.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp_lib, LIB)                          # same filesystem (objdir lives under the package): atomic
    tmp_stamp = stamp + ".%d.tmp" % os.getpid()
    with open(tmp_stamp, "w") as f:
        f.write(dig)
    os.replace(tmp_stamp, stamp)
    return LIB


def code_digest():
    """sha256 over csrc/*, include/*.h and the compile flags: identifies the kernel code a measurement belongs to (the GPU box has no .git)"""
    return _digest()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

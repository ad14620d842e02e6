"""SH helpers the renderer mirror needs (reference: shared_utils/sh_utils.py:26, 114-117).  The SH polynomial itself is
evaluated inside the HIP kernel (csrc/gs_math.h, basis as in sh_utils.py:57-100)."""
C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5

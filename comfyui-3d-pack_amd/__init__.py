"""ComfyUI custom-node entry of the MI355X-native MVs_Algorithms hot path.

Like the reference's __init__.py:12-14 this puts the package root on sys.path, which is also what makes the drop-in
packages `diff_gaussian_rasterization` and `nvdiffrast.torch` importable under their original names."""
import os
import sys

ROOT_PATH = os.path.dirname(os.path.abspath(__file__))
if ROOT_PATH not in sys.path:
    sys.path.insert(0, ROOT_PATH)

from nodes import NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS  # noqa: E402

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]

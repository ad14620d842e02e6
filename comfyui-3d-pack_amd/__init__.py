"""ComfyUI custom-node entry of the MI355X-native MVs_Algorithms hot path.

Like the reference's __init__.py:12-14 this puts the package root on sys.path, which is also what makes the drop-in
packages `diff_gaussian_rasterization` and `nvdiffrast.torch` importable under their original names."""
import os
import sys

import importlib

ROOT_PATH = os.path.dirname(os.path.abspath(__file__))
if ROOT_PATH not in sys.path:
    sys.path.append(ROOT_PATH)      # appended, as the reference does: never shadow the host application's own modules

# ComfyUI's own `nodes` module is what loads custom nodes, so it is already in sys.modules when this file runs: an absolute
# `from nodes import ...` would re-export ComfyUI's mappings.  Relative import, as the reference (__init__.py: importlib.import_module('.nodes', ...)).
if __package__:
    _nodes = importlib.import_module(".nodes", package=__name__)
else:                               # imported as a plain script directory (tests): load nodes.py by path under a private name
    import importlib.util as _ilu
    _spec = _ilu.spec_from_file_location("c3d_mi355x_nodes", os.path.join(ROOT_PATH, "nodes.py"))
    _nodes = _ilu.module_from_spec(_spec)
    _spec.loader.exec_module(_nodes)
NODE_CLASS_MAPPINGS, NODE_DISPLAY_NAME_MAPPINGS = _nodes.NODE_CLASS_MAPPINGS, _nodes.NODE_DISPLAY_NAME_MAPPINGS

__all__ = ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]

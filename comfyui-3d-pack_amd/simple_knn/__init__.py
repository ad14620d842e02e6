"""MI355X-native stand-in for the `simple_knn` wheel the reference imports (main_3DGS_renderer.py:408): `simple_knn._C.distCUDA2`."""

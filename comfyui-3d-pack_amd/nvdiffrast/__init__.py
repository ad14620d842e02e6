"""`nvdiffrast` -- drop-in package name for the MI355X-native mesh ops (see nvdiffrast/torch/__init__.py)."""
__version__ = "0.3.3+c3d.mi355x"

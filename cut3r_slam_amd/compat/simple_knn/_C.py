"""simple_knn._C: distCUDA2 (see the package docstring)"""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from cut3r_slam_amd.gaussian_rasterizer import distCUDA2  # noqa: E402,F401

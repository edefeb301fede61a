"""`diff_gaussian_rasterization` under the reference's import name (`from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer`: /root/reference/hislam2/gaussian/renderer/__init__.py:14).  The reference vendors a
CUDA extension (thirdparty/diff-gaussian-rasterization); these classes run the gfx950 rasteriser (`cut3r_gs_*`), forward and backward."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from cut3r_slam_amd.gaussian_rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: E402,F401
                                                rasterize_gaussians)

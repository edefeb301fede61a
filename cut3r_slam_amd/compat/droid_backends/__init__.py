"""`droid_backends` under the reference's import name (/root/reference/hislam2/modules/corr.py:4,12,19,79,87;
hislam2/geom/ba.py:167,200; hislam2/util/droid_visualization.py:97,100).  The reference's C++/CUDA sources for this module are
absent (setup.py:10-17 lists files that do not exist); the operators here are the gfx950 implementations."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from cut3r_slam_amd.droid_backends import *  # noqa: E402,F401,F403
from cut3r_slam_amd import droid_backends as _impl  # noqa: E402

globals().update({k: getattr(_impl, k) for k in dir(_impl) if not k.startswith("_")})

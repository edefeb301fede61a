"""`lietorch` under the reference's import name (`from lietorch import SE3`: /root/reference/hislam2/track_backend.py:6;
SO3 / Sim3: hislam2/gs_backend_per_frame.py:722-731, hislam2/geom/projective_ops.py, hislam2/pgo_buffer.py).  The reference's
submodule thirdparty/lietorch is empty; these classes run on the `cut3r_lie_*` HIP kernels with autograd."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from cut3r_slam_amd.lietorch import *  # noqa: E402,F401,F403
from cut3r_slam_amd.lietorch import SE3, SO3, Sim3  # noqa: E402,F401

"""`from src.dust3r.model import ARCroco3DStereo` (/root/reference/hislam2/hi2.py:5,21): the MI355X runtime of the CUT3R network behind
the reference class's name -- `from_pretrained(path)`, `.normalize`, `.encode_image`, `forward(views, ret_state)`, `.to()`, `.eval()`
(/root/reference/src/dust3r/model.py:305-318,1102-1114,894-900).  `ARCroco3DStereoOutput` is the (ress, views) result type (:50-56)."""
from . import _root  # noqa: F401  (puts the repository on sys.path)
from cut3r_slam_amd.model import ARCroco3DStereo, ARCroco3DStereoOutput, Cut3rModel  # noqa: E402,F401

__all__ = ["ARCroco3DStereo", "ARCroco3DStereoOutput", "Cut3rModel"]

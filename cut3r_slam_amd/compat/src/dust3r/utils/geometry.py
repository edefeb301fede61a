"""`from src.dust3r.utils.geometry import geotrf` (/root/reference/hislam2/track_frontend.py:11, track_backend.py:9;
/root/reference/src/dust3r/utils/geometry.py:49-124)."""
from .. import _root  # noqa: F401
from cut3r_slam_amd.dust3r_utils import geotrf, inv  # noqa: E402,F401

__all__ = ["geotrf", "inv"]

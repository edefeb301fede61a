"""`from src.dust3r.utils.camera import pose_encoding_to_camera` (/root/reference/hislam2/track_frontend.py:9, track_backend.py:8;
/root/reference/src/dust3r/utils/camera.py:364-420)."""
from .. import _root  # noqa: F401
from cut3r_slam_amd.dust3r_utils import pose_encoding_to_camera, quaternion_to_matrix  # noqa: E402,F401

__all__ = ["pose_encoding_to_camera", "quaternion_to_matrix"]

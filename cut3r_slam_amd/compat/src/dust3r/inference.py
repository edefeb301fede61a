"""`from src.dust3r.inference import inference` (/root/reference/hislam2/track_frontend.py:8, track_backend.py:7): same signature
and return value as /root/reference/src/dust3r/inference.py:219-239 -- `inference(groups, model, device, verbose=False) ->
(dict(views=..., pred=...), state_args)`, the view dicts are moved to `device` in place."""
from . import _root  # noqa: F401
from cut3r_slam_amd.inference import inference  # noqa: E402,F401

__all__ = ["inference"]

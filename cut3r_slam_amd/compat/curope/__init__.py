"""`curope` under the reference's import name (/root/reference/src/croco/models/curope/curope2d.py:7-10 does
`import curope as _kernels` and calls `_kernels.rope_2d(tokens, positions, base, F0)`): same signature, in-place semantics and
RuntimeErrors as curope.cpp:49-65 / kernels.cu:84-108, on the gfx950 kernel behind `cut3r_rope2d`."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from cut3r_slam_amd.ops import rope_2d  # noqa: E402,F401

__all__ = ["rope_2d"]

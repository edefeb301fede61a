"""`dust3r` -- the second spelling of the same package (the reference's model files import each other as `dust3r.*` after
/root/reference/src/dust3r/model.py:4 puts `src/` on sys.path).  Same objects as `src.dust3r`."""
import os
import sys

_here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _here not in sys.path:
    sys.path.insert(0, _here)

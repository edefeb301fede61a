from src.dust3r.utils.geometry import *  # noqa: F401,F403
from src.dust3r.utils.geometry import __all__  # noqa: F401

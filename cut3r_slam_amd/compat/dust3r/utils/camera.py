from src.dust3r.utils.camera import *  # noqa: F401,F403
from src.dust3r.utils.camera import __all__  # noqa: F401

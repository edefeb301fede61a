from src.dust3r.inference import *  # noqa: F401,F403
from src.dust3r.inference import __all__  # noqa: F401

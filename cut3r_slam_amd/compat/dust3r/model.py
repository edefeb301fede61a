from src.dust3r.model import *  # noqa: F401,F403
from src.dust3r.model import __all__  # noqa: F401

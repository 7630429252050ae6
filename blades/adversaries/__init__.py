from blades_amd.adversaries import *  # noqa: F401,F403
from blades_amd.adversaries import __all__  # noqa: F401

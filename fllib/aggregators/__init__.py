from blades_amd.aggregators import *  # noqa: F401,F403
from blades_amd.aggregators import __all__  # noqa: F401

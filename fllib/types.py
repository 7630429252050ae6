from blades_amd.types import *  # noqa: F401,F403
from blades_amd.types import NotProvided  # noqa: F401

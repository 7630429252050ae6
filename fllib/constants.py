from blades_amd.constants import *  # noqa: F401,F403

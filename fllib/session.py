from blades_amd.session import *  # noqa: F401,F403
from blades_amd.session import FLSession, get_current_worker, init_session, get_session, get_global_model, get_global_optimizer  # noqa: F401

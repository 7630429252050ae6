"""``fllib/utils/actor_pool.py`` surface; the implementation lives with the rank task farm."""
from blades_amd.parallel.actor_manager import ActorPool, _ActorPool  # noqa: F401

from blades_amd.parallel.actor_manager import ActorManager, _ActorPool, assign_to_ranks  # noqa: F401

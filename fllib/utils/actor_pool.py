from blades_amd.utils.actor_pool import ActorPool  # noqa: F401

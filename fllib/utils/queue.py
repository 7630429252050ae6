from blades_amd.utils.queue import Queue, Empty, Full  # noqa: F401

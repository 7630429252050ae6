from blades_amd.parallel.session import get_session  # noqa: F401

from blades_amd.algorithms import Algorithm, AlgorithmConfig, Server, ServerConfig, ClientManager  # noqa: F401

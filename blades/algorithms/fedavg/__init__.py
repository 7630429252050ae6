from blades_amd.algorithms.fedavg import Fedavg, FedavgConfig, FedavgDP, FedavgDPConfig  # noqa: F401

from .fedavg import Fedavg, FedavgConfig
from .fedavg_dp import FedavgDP, FedavgDPConfig

__all__ = ["Fedavg", "FedavgConfig", "FedavgDP", "FedavgDPConfig"]

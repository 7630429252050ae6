from .algorithm_config import AlgorithmConfig
from .algorithm import Algorithm
from .server import Server
from .server_config import ServerConfig
from .client_manager import ClientManager
from .lr_schedule import LearningRateSchedule, PiecewiseSchedule

__all__ = ["Algorithm", "AlgorithmConfig", "Server", "ServerConfig", "ClientManager", "LearningRateSchedule",
           "PiecewiseSchedule"]

from .callbacks import (ClientCallback, ClientCallbackList, ClippingCallback, LabelFlipCallback,
                        SignFlipCallback)
from .client_config import ClientConfig
from .client import BaseClient, Client, ClientProxy
from .dp_client import DPClient, DPCliengConfig, DPClientConfig

__all__ = ["ClientConfig", "BaseClient", "Client", "ClientProxy", "ClientCallback", "ClientCallbackList",
           "ClippingCallback", "SignFlipCallback", "LabelFlipCallback", "DPClient", "DPCliengConfig",
           "DPClientConfig"]

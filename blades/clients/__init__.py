from blades_amd.clients import Client, ClientProxy, ClippingCallback, DPClient, DPCliengConfig  # noqa: F401
from blades_amd.clients import callbacks, dp_client  # noqa: F401

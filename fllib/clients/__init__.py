from blades_amd.clients import BaseClient as Client  # the generic (non-malicious) client of fllib
from blades_amd.clients import ClientCallback, ClientCallbackList, ClientConfig  # noqa: F401

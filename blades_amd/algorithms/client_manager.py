"""ClientManager (reference ``fllib/algorithms/client_manager.py``): one client
object per dataset client id, order = dataset order."""
import copy
from typing import Dict, List

from blades_amd.utils.from_config import from_config


class ClientManager:
    def __init__(self, client_ids: List[str], client_config) -> None:
        self._clients = []
        self._by_id: Dict[str, object] = {}
        for uid in client_ids:
            cfg = copy.copy(client_config)  # shallow: per-client uid, shared callbacks_config spec
            cfg.uid = uid
            client = from_config(cfg.cls, {"client_config": cfg})
            self._clients.append(client)
            self._by_id[uid] = client

    @property
    def clients(self) -> list:
        return self._clients

    def get_client_by_id(self, uid):
        return self._by_id[uid]

    def __len__(self):
        return len(self._clients)

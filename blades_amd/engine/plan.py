"""Static client -> rank/row plan.

Rows of the global update matrix are laid out rank-major: rank 0's clients
first, then rank 1's, … .  Training clients and non-training (update-forging)
malicious clients are dealt to ranks separately and round-robin, so every rank
gets the same number of clients to train (±1) — non-training clients cost
nothing and must not unbalance the compute.
"""
from dataclasses import dataclass
from typing import List

import numpy as np


@dataclass
class RowPlan:
    world_size: int
    rank_clients: List[List[int]]  # per rank: global client indices, in row order
    row_of_client: np.ndarray  # [K] global row of each client
    client_of_row: np.ndarray  # [K]
    row_start: np.ndarray  # [G+1] first global row of each rank

    def local_clients(self, rank: int) -> List[int]:
        return self.rank_clients[rank]

    def rows_per_rank(self) -> List[int]:
        return [len(c) for c in self.rank_clients]


def make_plan(trains: List[bool], world_size: int) -> RowPlan:
    K = len(trains)
    rank_clients: List[List[int]] = [[] for _ in range(world_size)]
    if world_size == 1:
        rank_clients[0] = list(range(K))
    else:
        t = [i for i in range(K) if trains[i]]
        nt = [i for i in range(K) if not trains[i]]
        for pos, i in enumerate(t):
            rank_clients[pos % world_size].append(i)
        # deal non-training clients starting from the emptiest ranks
        order = np.argsort([len(c) for c in rank_clients], kind="stable")
        for pos, i in enumerate(nt):
            rank_clients[int(order[pos % world_size])].append(i)
        rank_clients = [sorted(c) for c in rank_clients]
    client_of_row = np.array([i for c in rank_clients for i in c], dtype=np.int64)
    row_of_client = np.empty(K, dtype=np.int64)
    row_of_client[client_of_row] = np.arange(K)
    row_start = np.concatenate([[0], np.cumsum([len(c) for c in rank_clients])]).astype(np.int64)
    return RowPlan(world_size, rank_clients, row_of_client, client_of_row, row_start)

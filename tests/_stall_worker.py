"""Rank 1 marks progress once and then hangs; rank 0 keeps beating (tests/test_launcher.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blades_amd.parallel.runtime import heartbeat

rank = int(os.environ["RANK"])
heartbeat("round 0")
for i in range(600):
    if rank == 0:
        heartbeat(f"round {i}")
    time.sleep(0.1)

"""Framework-wide constants.

Parity: reference ``fllib/constants.py:17-42`` (same names and values so that
result dicts / user code written against the reference keep working).
"""
import os

CLIENT_UPDATE = "client_update"
GLOBAL_MODEL = "global_model"
NUM_GLOBAL_STEPS = "num_global_steps"
MAIN_ACTOR = "main_actor"

DEFAULT_DATA_ROOT = "./fldata"
DEFAULT_RESULTS_ROOT = "blades_results"

SERVER_RANK = 0
DEFAULT_MASTER_ADDR = "127.0.0.1"
DEFAULT_MASTER_PORT = "22225"
MINIMUM_GPU_FRACTION = 1e-4

# Registry categories (the reference piggybacks on Tune's global registry).
FLLIB_DATASET = "fllib_dataset"
FLLIB_MODEL = "fllib_model"

# Column padding of the sharded update matrix (elements). 256 fp32 = 1 KiB, one
# wave64 x float4 access, so every rank's column slice starts 1 KiB aligned.
COL_ALIGN = 256


def env_integer(key: str, default: int) -> int:
    """Integer environment override (reference ``constants.py:5-14``)."""
    if key in os.environ:
        val = os.environ[key]
        if val == "inf":
            return 2 ** 63 - 1
        return int(val)
    return default


COMMUNICATOR_INTERVAL_S = env_integer("COMMUNICATOR_INTERVAL_S", 1)

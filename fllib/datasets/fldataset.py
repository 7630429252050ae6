from blades_amd.datasets.fldataset import *  # noqa: F401,F403
from blades_amd.datasets.fldataset import FLDataset, CustomTensorDataset, set_random_seed  # noqa: F401

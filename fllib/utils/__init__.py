from blades_amd.utils import torch_utils  # noqa: F401

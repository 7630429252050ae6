from blades_amd.utils.torch_utils import *  # noqa: F401,F403
from blades_amd.utils.torch_utils import clip_tensor_norm_, clip_para_norm_, parameters_to_vector, vector_to_parameters, set_torch_seed  # noqa: F401

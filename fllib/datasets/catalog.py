from blades_amd.datasets.catalog import *  # noqa: F401,F403
from blades_amd.datasets.catalog import DatasetCatalog, make_dataset  # noqa: F401

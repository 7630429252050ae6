"""blades_amd — MI355X-native Byzantine-robust federated-learning simulator."""
import os as _os

# MIOpen's reference ("naive") direct-convolution solvers are only useful as a correctness
# fallback; on gfx950 its find step still benchmarks them (60-200 ms per call on the batched
# shapes used here, tens of seconds per process — see profiles/r1_v0_vmap_engine_kernel_stats.csv)
# and its immediate mode may even select them.  Exclude them unless the user decided otherwise.
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    _os.environ.setdefault(_k, "0")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"

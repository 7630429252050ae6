"""Row-by-row deviation of the fused per-client-weights path (3 local steps, momentum 0.9, ResNet-10) from the CPU
vmap oracle; run under different BLADES_CHUNK_STREAMS / chunk sizes to separate ReLU-gate events from races:
python tools/probes/local_steps_rows_probe.py <clients_per_chunk> <out.pt>"""
import copy, sys, tempfile
import torch
sys.path.insert(0, ".")
from tests.test_engine_paths import build, one_round_updates

chunk, out = int(sys.argv[1]), sys.argv[2]
root = tempfile.mkdtemp()
b = build(root, "resnet", "vmap", "cifar10", nb=3, momentum=0.9, lr=0.02)
cfg = copy.deepcopy(b.config); cfg._is_frozen = False
cfg.device, cfg.amp_dtype, cfg.execution_mode, cfg.clients_per_chunk = "cuda", "fp32", "fused", chunk
g = cfg.build()
g.server.theta.copy_(b.server.theta.cuda())
Ug, _ = one_round_updates(g)
Ub, _ = one_round_updates(b)
rel = (Ug.cpu() - Ub).norm(dim=1) / Ub.norm(dim=1)
print("chunk", chunk, "rel", [f"{v:.2e}" for v in rel.tolist()], flush=True)
torch.save(Ug.cpu(), out)

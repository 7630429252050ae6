"""Row-by-row error of one fused ResNet-10 round on the GPU against the fp32 CPU oracle, per compute mode."""
import copy, sys, tempfile
import torch
sys.path.insert(0, ".")
from tests.test_engine_paths import build, one_round_updates

root = tempfile.mkdtemp()
b = build(root, "resnet", "fused")
theta0 = b.server.theta.clone()
Ub, _ = one_round_updates(b)
# a second CPU evaluation in float64 would be the true oracle; use vmap path as an independent fp32 implementation
v = build(root, "resnet", "vmap")
v.server.theta.copy_(theta0)
Uv, _ = one_round_updates(v)
def report(name, U):
    d = (U - Ub)
    rel = d.norm(dim=1) / Ub.norm(dim=1)
    bad = (d.abs() > 1e-5 + 1e-3 * Ub.abs())
    print(f"{name}: row rel L2 {rel.tolist()}  mismatched {int(bad.sum())}/{bad.numel()}  max abs {float(d.abs().max()):.3e}  |U|max {float(Ub.abs().max()):.3e}", flush=True)
report("cpu vmap fp32", Uv)
for amp in ("fp32", "bf16x3", "bf16"):
    cfg = copy.deepcopy(b.config); cfg._is_frozen = False
    cfg.device, cfg.amp_dtype = "cuda", amp
    g = cfg.build()
    g.server.theta.copy_(theta0.cuda())
    Ug, _ = one_round_updates(g)
    report("gpu " + amp, Ug.cpu())
import os
os.environ["BLADES_CONV_IMPL"] = "miopen"

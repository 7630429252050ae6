"""A few Gram launches at small client counts for a PMC pass (tools/r3/run57.sh)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from blades_amd import ops
h = ops._h()
dev = torch.device('cuda')
for K, D in ((60, 4_903_242), (128, 4_903_242), (60, 1_475_146)):
    U = torch.randn(K, (D + 255) // 256 * 256, device=dev)[:, :D]
    for _ in range(3):
        h.gram(U)
    m = torch.randn(D, device=dev)
    for _ in range(2):
        ops.sqdist_to(U, m)
torch.cuda.synchronize()

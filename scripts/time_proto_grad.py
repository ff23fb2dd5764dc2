"""get_pred forward + both gradients at B=32 x 2048, C=128, K=14 (shared and per-sample prototypes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d import ops
dev = torch.device("cuda:0")
f = torch.randn(32, 128, 2048, device=dev).requires_grad_(True)
for shape in ((14, 128), (32, 14, 128)):
    p = torch.randn(*shape, device=dev).requires_grad_(True)
    cot = torch.randn(32, 14, 2048, device=dev)
    for _ in range(12):
        f.grad = p.grad = None
        ops.ProtoCosineFunction.apply(f, p, 10.0).backward(cot)
torch.cuda.synchronize()

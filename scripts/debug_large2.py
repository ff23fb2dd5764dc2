import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from gfs3d import ops
from oracle import ref_ops as R
from test_gpu_parity import _block_params, _make_block
dev = torch.device("cuda:0")
B, C, N, k = 16, 9, 2048, 20
p = _block_params(C, 555)
g = torch.Generator().manual_seed(99)
x = torch.randn(B, C, N, generator=g); cot = torch.randn(B, 64, N, generator=g)
idx = ops.knn_topk(x.to(dev), k).cpu()
blk = _make_block(p, C, dev).train()
c1, bn1, c2, bn2 = blk.fused_parts()
for trial in range(3):
    for q in blk.parameters(): q.grad = None
    xg = x.to(dev).requires_grad_(True)
    y = ops.edgeconv_block(xg, idx.int().to(dev), c1.weight, bn1, c2.weight, bn2, True)
    (y * cot.to(dev)).sum().backward()
    gd = (cot.to(dev) * torch.where(y.detach() > 0, 1.0, 0.2)).double()
    db2_t = gd.sum(dim=(0, 2))
    print("trial", trial, "db2 vs torch-from-y:", float((bn2.bias.grad.double() - db2_t).abs().max() / db2_t.abs().max()),
          " |db2|max", float(db2_t.abs().max()), " dx norm", float(xg.grad.norm()))

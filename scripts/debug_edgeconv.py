"""Debug helper (GPU): per-quantity errors of the HIP EdgeConv block vs torch autograd."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from gfs3d import ops
from oracle import c_oracle, ref_ops as R
from test_gpu_parity import _block_params, _make_block

dev = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))

for (B, C, N, k) in [(2, 9, 256, 20), (2, 64, 128, 20), (1, 9, 131, 7), (1, 9, 256, 20), (2, 9, 128, 20), (2, 9, 256, 7), (4, 64, 256, 20)]:
    p = _block_params(C, 200 + C)
    g = torch.Generator().manual_seed(N + k)
    x = torch.randn(B, C, N, generator=g); cot = torch.randn(B, 64, N, generator=g)
    idx = torch.from_numpy(c_oracle.knn(x.numpy(), k))
    pr = {kk: v.clone().double().requires_grad_("running" not in kk) for kk, v in p.items()}
    xr = x.clone().double().requires_grad_(True)
    yr = R.conv_stack(R.edge_feature(xr, k, idx), pr, "b", 2, training=True, new_stats={}).max(dim=-1)[0]
    (yr * cot.double()).sum().backward()
    blk = _make_block(p, C, dev).train()
    c1, bn1, c2, bn2 = blk.fused_parts()
    xg = x.to(dev).requires_grad_(True)
    y = ops.edgeconv_block(xg, idx.int().to(dev), c1.weight, bn1, c2.weight, bn2, True)
    (y * cot.to(dev)).sum().backward()
    print(f"B={B} C={C} N={N} k={k}: y {rel(y, yr):.2e} dx {rel(xg.grad, xr.grad):.2e} dw1 {rel(c1.weight.grad, pr['b.layer.0.weight'].grad):.2e} "
          f"dw2 {rel(c2.weight.grad, pr['b.layer.3.weight'].grad):.2e} dg1 {rel(bn1.weight.grad, pr['b.layer.1.weight'].grad):.2e} "
          f"db1 {rel(bn1.bias.grad, pr['b.layer.1.bias'].grad):.2e} dg2 {rel(bn2.weight.grad, pr['b.layer.4.weight'].grad):.2e} "
          f"db2 {rel(bn2.bias.grad, pr['b.layer.4.bias'].grad):.2e}")
    # where is dx wrong?
    d = (xg.grad.detach().cpu().double() - xr.grad).abs()
    bad = (d > 1e-3 * xr.grad.abs().max()).nonzero()
    print("   bad dx entries:", bad.shape[0], bad[:6].tolist())

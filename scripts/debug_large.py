import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
from gfs3d import ops
from oracle import ref_ops as R
from test_gpu_parity import _block_params, _make_block, _force_routing
dev = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
for (B, C, N, k, dbl) in [(16, 64, 2048, 20, False), (16, 64, 2048, 20, True), (4, 64, 2048, 20, True), (16, 9, 2048, 20, True)]:
    p = _block_params(C, 555)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, C, N, generator=g); cot = torch.randn(B, 64, N, generator=g)
    idx = ops.knn_topk(x.to(dev), k).cpu()
    deg = torch.bincount((idx + torch.arange(B)[:, None, None] * N).flatten(), minlength=B * N)
    dt = torch.float64 if dbl else torch.float32
    pr = {kk: v.clone().to(dt).requires_grad_("running" not in kk) for kk, v in p.items()}
    xr = x.clone().to(dt).requires_grad_(True)
    yr, jr = R.conv_stack(R.edge_feature(xr, k, idx), pr, "b", 2, training=True, new_stats={}).max(dim=-1)
    (yr * cot.to(dt)).sum().backward()
    blk = _make_block(p, C, dev).train()
    c1, bn1, c2, bn2 = blk.fused_parts()
    xg = x.to(dev).requires_grad_(True)
    y = ops.edgeconv_block(xg, idx.int().to(dev), c1.weight, bn1, c2.weight, bn2, True)
    fl = _force_routing(y, torch.gather(idx.unsqueeze(1).expand(B, 64, N, k), 3, jr.unsqueeze(-1)).squeeze(-1))
    (y * cot.to(dev)).sum().backward()
    print(f"B={B} C={C} ref={'fp64' if dbl else 'fp32'} maxdeg={int(deg.max())} flips={fl}: y {rel(y, yr):.1e} dx {rel(xg.grad, xr.grad):.1e} dw1 {rel(c1.weight.grad, pr['b.layer.0.weight'].grad):.1e} "
          f"dw2 {rel(c2.weight.grad, pr['b.layer.3.weight'].grad):.1e} dg1 {rel(bn1.weight.grad, pr['b.layer.1.weight'].grad):.1e} db1 {rel(bn1.bias.grad, pr['b.layer.1.bias'].grad):.1e} "
          f"dg2 {rel(bn2.weight.grad, pr['b.layer.4.weight'].grad):.1e} db2 {rel(bn2.bias.grad, pr['b.layer.4.bias'].grad):.1e}")

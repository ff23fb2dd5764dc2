import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import torch.nn.functional as F
from gfs3d import ops
from oracle import c_oracle, ref_ops as R
from test_gpu_parity import _block_params, _make_block
dev = torch.device("cuda:0")
B, C, N, k = 2, 9, 256, 20
p = _block_params(C, 200 + C)
g = torch.Generator().manual_seed(N + k)
x = torch.randn(B, C, N, generator=g); cot = torch.randn(B, 64, N, generator=g)
idx = torch.from_numpy(c_oracle.knn(x.numpy(), k))
pr = {kk: v.clone().double().requires_grad_("running" not in kk) for kk, v in p.items()}
xr = x.clone().double().requires_grad_(True)
e = R.edge_feature(xr, k, idx)
h1 = F.conv2d(e, pr["b.layer.0.weight"])
z1 = F.batch_norm(h1, None, None, pr["b.layer.1.weight"], pr["b.layer.1.bias"], True, 0.1, 1e-5); z1.retain_grad()
a1 = F.leaky_relu(z1, 0.2)
h2 = F.conv2d(a1, pr["b.layer.3.weight"])
z2 = F.batch_norm(h2, None, None, pr["b.layer.4.weight"], pr["b.layer.4.bias"], True, 0.1, 1e-5)
yr, jr = F.leaky_relu(z2, 0.2).max(dim=-1)
(yr * cot.double()).sum().backward()
GQ_ref = z1.grad.sum(-1).permute(0, 2, 1)      # (B,N,64)
blk = _make_block(p, C, dev).train()
c1, bn1, c2, bn2 = blk.fused_parts()
xg = x.to(dev).requires_grad_(True)
y = ops.edgeconv_block(xg, idx.int().to(dev), c1.weight, bn1, c2.weight, bn2, True)
ctx = y.grad_fn.c
jsel = ctx.jsel.cpu().long()                   # (B,N,64)
print("jsel mismatch:", int((jsel != jr.permute(0, 2, 1)).sum()), "of", jsel.numel())
mm = (jsel != jr.permute(0, 2, 1)).nonzero()
print(mm[:10].tolist())
for (b, n, c) in mm[:5].tolist():
    print("  h2 row:", h2[b, c, n].detach().numpy().round(5), "ref j", int(jr[b, c, n]), "mine", int(jsel[b, n, c]), "z2row", z2[b,c,n].detach().numpy().round(5))
(y * cot.to(dev)).sum().backward()
ws = list(ops._workspaces.values())[0]
BN = B * N
def al(v): return (v + 255) // 256 * 256
tiles = ((N + 63) // 64) * B; gh = min(1024, (BN + 7) // 8); mp = max(tiles, gh)
o = 0
def take(nb):
    global o
    r = o; o = al(o + nb); return r
off_part = take(8 * 2 * 64 * mp); off_hmax = take(4 * BN * 64); off_hmin = take(4 * BN * 64); off_jmax = take(BN * 64); off_jmin = take(BN * 64)
off_g = take(4 * BN * 64); off_GP = take(4 * BN * 64); off_GQ = take(4 * BN * 64)
GQ = ws[off_GQ:off_GQ + 4 * BN * 64].view(torch.float32).view(B, N, 64).cpu().double()
err = (GQ - GQ_ref).abs().amax(-1)     # (B,N)
print("GQ max err per point: worst", err.max().item(), "ref max", GQ_ref.abs().max().item())
bad = (err > 1e-4 * GQ_ref.abs().max()).nonzero()
print("bad source points:", bad.tolist()[:20])
for (b, n) in bad.tolist()[:3]:
    print(" point", b, n, "idx row", idx[b, n].tolist())
    print("   GQ mine", GQ[b, n, :6].numpy(), "ref", GQ_ref[b, n, :6].detach().numpy())

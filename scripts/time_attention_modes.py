import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import torch
from gfs3d import ops
from test_gpu_reduced import cosine
dev = torch.device("cuda:0")
for scale in (1.0, 2.0, 3.0):
    B, N = 2, 512
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(B, 64, N, generator=g) * s for s in (scale, scale, 1.0))
    cot = torch.randn(B, 64, N, generator=g)
    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    att = torch.softmax(torch.bmm(q64.transpose(1, 2), k64) / 8.0, dim=-1)
    torch.bmm(att, v64.transpose(1, 2)).transpose(1, 2).backward(cot.double())
    for mode in ("f32", "f16", "bf16"):
        qd, kd, vd = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        with ops.compute(mode):
            ops.attention(qd, kd, vd, 1.0 / 8.0, 0.0, 0, None).backward(cot.to(dev))
        print(os.environ.get("GFS3D_LIB", "new")[-16:], scale, mode, ["%.2e" % (1.0 - cosine(a_.grad, b_.grad)) for a_, b_ in ((qd, q64), (kd, k64), (vd, v64))])

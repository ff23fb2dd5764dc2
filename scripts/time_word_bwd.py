"""Time the word-assignment backward (proto_bwd_x6_kernel) at B=32 x 2048, C=192, K_w=200 (device events)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
from gfs3d import ops
dev = torch.device("cuda:0")
B, N = 32, 2048
g = torch.Generator().manual_seed(0)
feat = torch.randn(B, 192, N, generator=g).to(dev); gp = torch.randn(200, 192, generator=g).to(dev)
soft, _, _ = ops.word_assign_raw(feat, gp, 10.0)
dsoft = torch.randn(B, 200, N, generator=g).to(dev)
fn = lambda: ops.proto_backward_feat(feat, gp, 10.0, dsoft, soft)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("GFS3D_LIB", "shipped")[-22:], "word-assign backward: %.4f ms" % (e0.elapsed_time(e1) / 20))

"""Time the fused prototype head (csrc/head.hip) at configs[3] / configs[4] shapes; run under scripts/prof_one.sh for per-kernel times."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
from gfs3d import ops
dev = torch.device("cuda:0")
for B, classes, base, fake in ((32, 13, 7, [2, 4, 7]), (64, 21, 15, [1, 3, 5, 8, 9, 12, 14])):
    g = torch.Generator().manual_seed(0)
    f = torch.randn(B, 128, 2048, generator=g).to(dev).requires_grad_(True)
    mp = torch.randn(classes, 128, generator=g).to(dev).requires_grad_(True)
    bg = torch.randn(1, 128, generator=g).to(dev).requires_grad_(True)
    y = torch.randint(0, base + 1, (B, 2048), generator=g).to(dev)
    def step():
        f.grad = mp.grad = bg.grad = None
        ops.capl_train_head(f, mp, bg, y, fake, base, 255)[1].backward()
    for _ in range(50): step()          # (a short run measures the clock ramp of a cold device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): step()
    e1.record(); torch.cuda.synchronize()
    print(f"train head fwd+bwd B={B} classes={classes}: {e0.elapsed_time(e1) / 200:.4f} ms")

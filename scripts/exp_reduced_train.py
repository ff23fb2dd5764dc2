"""Loss curves of a short CAPL training run in the three arithmetic modes from one seed (configs[4] shapes: 21 + 1 classes,
K_w = 180): does a reduced mode TRAIN like fp32?  python scripts/exp_reduced_train.py [steps] [B] [N]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import contextlib, io
import torch
import bench
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks


def run(mode, steps, B, N, classes=21, base=15, Kw=180, lr=1e-3, dev=torch.device("cuda:0")):
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    torch.manual_seed(0)
    gp = torch.randn(Kw, 192, generator=torch.Generator().manual_seed(2)).to(dev)
    with contextlib.redirect_stdout(io.StringIO()):
        net = mpti_net_Point_GeoAsWeight_v2(classes=classes, args=bench.capl_args(), base_num=base, gp=gp, energy=0.9)
    net = net.to(dev).train()
    x = synthetic_blocks(B, N, seed=3).to(dev)
    # learnable labels: 0 (background) .. base by height and colour, so that the loss can fall
    y = ((x[:, 2] * 0.999 * 4).long() * 4 + (x[:, 3] * 0.999 * 4).long()).clamp(0, base).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    losses = []
    torch.manual_seed(1)
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        random.seed(s)
        with ops.compute(mode):
            _, loss = net(x, y=y)
            loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    curves = {m: run(m, steps, B, N) for m in ("f32", "f32", "f16", "bf16")} if False else {}
    ref = run("f32", steps, B, N)
    ref2 = run("f32", steps, B, N)
    h = run("f16", steps, B, N)
    b = run("bf16", steps, B, N)
    print("step   f32      f32(again) f16      bf16")
    for s in range(0, steps, max(1, steps // 20)):
        print(f"{s:4d}  {ref[s]:.5f}  {ref2[s]:.5f}  {h[s]:.5f}  {b[s]:.5f}")
    import numpy as np
    r = np.array(ref)
    for name, c in (("f32 again", ref2), ("f16", h), ("bf16", b)):
        c = np.array(c)
        d = np.abs(c - r) / r
        top = np.argsort(-d)[:4]
        print(f"{name}: max |dloss| / loss = {d.max():.4f} (step {int(d.argmax())}; next {[(int(t), round(float(d[t]), 4)) for t in top[1:]]}; "
              f"95th pct {np.percentile(d, 95):.4f}); last-10 mean {c[-10:].mean():.5f} vs {r[-10:].mean():.5f}")

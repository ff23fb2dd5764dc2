"""Probe: do two independent point-wise stack backward passes on two streams finish faster than back to back?  (Would a side stream for
the weight-gradient kernels of K4 pay?)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
from gfs3d import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def make():
    xs = [torch.randn(32, 2048, 64, generator=g).to(dev) for _ in range(3)]
    layers, prev = [], 192
    for c in (512, 256):
        layers.append(((torch.randn(c, prev, generator=g) / prev ** 0.5).to(dev), torch.ones(c, device=dev), torch.zeros(c, device=dev),
                       torch.zeros(c, device=dev), torch.ones(c, device=dev)))
        prev = c
    run = ops.PointwiseRun(xs, layers, True, 1e-5, 0.1, 0.2, private_workspace=True)
    run.forward()
    return run


a, b = make(), make()
dy = torch.randn(32, 256, 2048, generator=g).to(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def seq(n):
    t0 = time.perf_counter()
    for _ in range(n):
        a.backward(dy); b.backward(dy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def par(n):
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(s1):
            a.backward(dy)
        with torch.cuda.stream(s2):
            b.backward(dy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


seq(3); par(3)
for _ in range(3):
    print("two backward passes: back to back %.3f ms, on two streams %.3f ms" % (seq(20), par(20)))

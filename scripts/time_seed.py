"""Greedy k-means++ seeding alone at configs[2] size (2.1 M x 192, K = 200) - run under scripts/prof_one.sh for per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import numpy as np
import torch
from get_basis import kmeans_plus_plus

dev = torch.device("cuda:0")
M, C, K = 7 * 300000, 192, 200
g = torch.Generator().manual_seed(0)
pts = torch.randn(M, C, generator=g).to(dev)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kmeans_plus_plus(pts, K, np.random.RandomState(rep))
    torch.cuda.synchronize()
    print(f"greedy seeding: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)

"""configs[2] pieces at full size: k-means++ seeding (pruned / unpruned / the host form of rounds 1-3) on 2.1 M x 192 points, and the
whole extract_geometric_words flow with its stage split (bench.py: extra.c3_end_to_end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import numpy as np
import torch
from gfs3d import ops
import bench

dev = torch.device("cuda:0")
M, C, K = 7 * 300000, 192, 200
g = torch.Generator().manual_seed(0)
for name, pts in (("gaussian", torch.randn(M, C, generator=g)),
                  ("clustered", torch.randn(50, C, generator=g)[torch.randint(0, 50, (M,), generator=g)] + 0.3 * torch.randn(M, C, generator=g))):
    pts = pts.to(dev)
    rng = np.random.RandomState(0)
    first, u = int(rng.randint(M)), rng.random_sample(K - 1)
    for prune in (True, False):
        ops.kmeanspp_seed(pts[:100000], K, 5, u, prune=prune)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c, p = ops.kmeanspp_seed(pts, K, first, u, prune=prune)
        torch.cuda.synchronize()
        print(f"{name}: k-means++ seeding, prune={prune}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    if "--host" in sys.argv:
        from get_basis import kmeans_plus_plus_host
        t0 = time.perf_counter()
        kmeans_plus_plus_host(pts, K, np.random.RandomState(0))
        torch.cuda.synchronize()
        print(f"{name}: host form: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del pts
print(bench.c3_end_to_end(dev))

"""One Lloyd step (assignment + centroid sums) on BASELINE configs[2]'s 2.1 M x 192 points, 200 centroids; run under scripts/prof_one.sh."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
from gfs3d import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 7 * 300000           # (300000: the points fit the 256-MB memory-side cache)
pts = torch.randn(M, 192, generator=g).to(dev)
cent = pts[torch.randperm(M, generator=g)[:200].to(dev)].contiguous()
for fast in (False, True):
    for _ in range(2): ops.kmeans_assign(pts, cent, fast=fast)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): ops.kmeans_assign(pts, cent, fast=fast)
    torch.cuda.synchronize(); print("lloyd step ms", "screened" if fast else "exact", round((time.time() - t) / 5 * 1e3, 3))

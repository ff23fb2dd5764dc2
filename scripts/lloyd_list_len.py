"""How long is the undecided list of the screened Lloyd assignment?  Runs the c3 flow's k-means on its real EdgeConv features with a
hook that reads the list counter after every screened assignment.  python scripts/lloyd_list_len.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch, bench
from gfs3d import ops
lens = []
_orig = ops.kmeans_assign
def hooked(pts, cent, fast=False, want_sums=True):
    r = _orig(pts, cent, fast=fast, want_sums=want_sums)
    if fast:
        M = pts.shape[0]
        ws = ops._workspace(pts.device, 0)
        off = (4 * M + 255) // 256 * 256
        lens.append(int(ws[off:off + 4].view(torch.int32).item()))
    return r
ops.kmeans_assign = hooked
import get_basis
get_basis.ops.kmeans_assign = hooked
d = bench.c3_end_to_end(torch.device("cuda:0"))
import numpy as np
l = np.array(lens[-300:])
print("assignments", len(lens), "list length: first 10", l[:10].tolist(), "median", int(np.median(l)), "max", int(l.max()), "last 5", l[-5:].tolist())

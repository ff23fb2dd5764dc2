import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks
dev = torch.device("cuda:0")
x9 = synthetic_blocks(32, 2048, seed=0).to(dev)
x64 = torch.randn(32, 64, 2048, device=dev)
for name, x in (("C=9", x9), ("C=64", x64)):
    for _ in range(3): ops.knn_topk(x, 20, want_i64=False, want_i32=True)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): ops.knn_topk(x, 20, want_i64=False, want_i32=True)
    torch.cuda.synchronize(); print(name, "ms per call:", round((time.time() - t) / 10 * 1e3, 4))

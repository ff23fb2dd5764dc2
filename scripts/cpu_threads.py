import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from oracle import ref_ops as R
from model.dgcnn import DGCNN
torch.manual_seed(0)
sd = {k: v.clone() for k, v in DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).state_dict().items()}
x = R.synthetic_blocks(8, 2048, seed=0)
def one():
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    _, out = R.dgcnn_forward(p, x, k=20, training=True, new_stats={})
    out.mean().backward()
print("cpu count", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt); one(); t = time.time(); one(); one(); dt = (time.time() - t) / 2
    print(nt, "threads:", round(8 / dt, 2), "blocks/s")

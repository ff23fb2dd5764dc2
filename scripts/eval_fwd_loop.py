"""The three kNN + EdgeConv blocks alone, eval mode, B = 32 x 2048 (bench.py's `fwd_knn_edgeconv_eval`), 30 times - the
workload of a kernel-trace / timeline of the forward path:
  rocprofv3 --kernel-trace -d <dir> -o r -- python3 scripts/eval_fwd_loop.py && python3 scripts/timeline.py <dir>/r_results.db knn_prep_kernel 4 dump"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = synthetic_blocks(32, 2048, seed=0).to(dev)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev).eval()
with torch.no_grad():
    for _ in range(30):
        net.edgeconv_features(x)
torch.cuda.synchronize()

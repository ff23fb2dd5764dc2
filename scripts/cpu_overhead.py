import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d import dist as gdist
from model.dgcnn import DGCNN
from oracle.ref_ops import synthetic_blocks
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev).train()
bucket = gdist.FlatGradBucket(net.parameters()); opt = torch.optim.Adam(net.parameters(), lr=1e-3)
x = synthetic_blocks(32, 2048, seed=0).to(dev)
def step():
    bucket.zero(); _, out = net(x); out.mean().backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): step()
t_enq = time.perf_counter() - t
torch.cuda.synchronize(); t_all = time.perf_counter() - t
print("CPU enqueue ms/step:", round(t_enq / 20 * 1e3, 3), " wall ms/step:", round(t_all / 20 * 1e3, 3))

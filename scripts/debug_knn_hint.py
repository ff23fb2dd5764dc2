"""How tight is 'the 20th best distance among the PREVIOUS layer's neighbours' as a threshold for the next layer's kNN?
Prints, per EdgeConv layer >= 2, the mean / median / p90 number of candidates above that bound (20 = perfect)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
import torch
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=20, return_edgeconvs=True).to(dev).train()
x = synthetic_blocks(4, 2048, seed=0).to(dev)
with torch.no_grad():
    feats, _ = net(x)
    prev_idx = ops.knn_topk(x, 20).long()
    for l, f in enumerate(feats[:2]):
        ft = f.transpose(1, 2)                                   # (B,N,64)
        d = -torch.cdist(ft.double(), ft.double()) ** 2          # larger = nearer
        hint = torch.gather(d, 2, prev_idx)                      # (B,N,20)
        bound = hint.min(dim=2, keepdim=True)[0]
        above = (d > bound).sum(dim=2).float() + 1
        print(f"layer {l + 2}: candidates above the hint bound: mean {above.mean():.1f} median {above.median():.0f} "
              f"p90 {above.quantile(0.9):.0f} max {above.max():.0f}")
        prev_idx = ops.knn_topk(f.contiguous(), 20).long()

"""Would Hamerly's bounds pay in the Lloyd loop of configs[2]?  Per iteration: the share of points whose bounds (upper bound to the
assigned centre, lower bound to the second closest, both carried across the centre movements) do NOT prove that the label stays -
the points an accelerated step would have to re-score.  EdgeConv features of synthetic blocks, K = 200, sklearn's seeding."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import numpy as np
import torch
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks
from get_basis import DGCNNSeg, collect_edgeconv_features, subsample_per_class, kmeans_plus_plus

dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20)
torch.manual_seed(0)
seg = DGCNNSeg(a, num_classes=8).to(dev).eval()
blocks = torch.cat([synthetic_blocks(100, 2048, seed=100 + i) for i in range(nb // 100)]).to(dev)
labels = torch.randint(0, 8, (blocks.shape[0], 2048), generator=torch.Generator().manual_seed(1)).to(dev)
with torch.no_grad():
    feat = collect_edgeconv_features(seg, blocks, 64)
    pts = subsample_per_class(feat, labels, 8, 300000, 0).contiguous()
pts = pts - pts.mean(0)
M, K = pts.shape[0], 200
cent = kmeans_plus_plus(pts, K, np.random.RandomState(0)).contiguous()
print("points", M)


def two_best(p, c):
    d = torch.cdist(p, c)
    v, i = torch.topk(d, 2, dim=1, largest=False)
    return i[:, 0], v[:, 0], v[:, 1]


lab = ub = lb = None
for it in range(1, 81):
    if lab is None:
        parts = [two_best(pts[s:s + 65536], cent) for s in range(0, M, 65536)]
        lab, ub, lb = (torch.cat([p[j] for p in parts]) for j in range(3))
        share = 1.0
    else:
        move = (cent - prev).norm(dim=1)
        top = torch.topk(move, 2).values
        amax = int(move.argmax())
        ub = ub + move[lab]
        lb = lb - torch.where(lab == amax, top[1], top[0])
        cc = torch.cdist(cent, cent)
        cc.fill_diagonal_(float("inf"))
        shalf = 0.5 * cc.min(dim=1).values
        need = ub > torch.maximum(lb, shalf[lab])
        idx = torch.nonzero(need).flatten()
        share = idx.numel() / M
        if idx.numel():
            l2, u2, b2 = two_best(pts[idx], cent)
            lab[idx], ub[idx], lb[idx] = l2, u2, b2
    sums = torch.zeros(K, pts.shape[1], device=dev).index_add_(0, lab, pts)
    cnt = torch.bincount(lab, minlength=K).float()
    prev = cent
    cent = torch.where(cnt[:, None] > 0, sums / cnt.clamp_min(1)[:, None], cent)
    if it <= 12 or it % 10 == 0:
        print(f"iteration {it:3d}: re-scored share {share:.4f}  max move {float((cent - prev).norm(dim=1).max()):.5f}", flush=True)

"""VERDICT r4 item 2, measured before building: would Morton-ordering the points of a block make the kNN graphs LOCAL in index space
(neighbouring queries sharing candidates, a super-tile's edges sharing targets, dy1 segments staying in one workgroup)?
For the bench's synthetic blocks (uniform xyz, independent colours: SURVEY 8d recipe) and for a structured cloud (points on a few
planes with piecewise-constant colour, like a room), per EdgeConv layer of a seeded DGCNN:
  near    share of the k neighbours whose rank lies within +-W of the query's rank, in the given order and in Morton order
  targets distinct neighbour ids over the 8 queries of a super-tile (8 k = 160 edges) / over a 64-query tile (1280 edges)
  windows distinct 64-row windows (16 KB of P' rows) the 160 edges of a super-tile touch
Run on the GPU box: python scripts/exp_morton_locality.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd"), os.path.join(ROOT, "scripts")):
    sys.path.insert(0, _p)
import torch
from gfs3d import ops
from gfs3d.synthetic import synthetic_blocks
from model.dgcnn import DGCNN

dev = torch.device("cuda:0")
B, N, K = 8, 2048, 20


def morton_rank(xyz):
    """(B,3,N) -> (B,N) rank of every point along the Morton curve of its block (10 bits per axis)."""
    lo, hi = xyz.amin(dim=2, keepdim=True), xyz.amax(dim=2, keepdim=True)
    q = ((xyz - lo) / (hi - lo).clamp_min(1e-9) * 1023).long().clamp_(0, 1023)
    code = torch.zeros(xyz.shape[0], xyz.shape[2], dtype=torch.long, device=xyz.device)
    for bit in range(10):
        for ax in range(3):
            code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
    order = code.argsort(dim=1)
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(xyz.shape[2], device=xyz.device).expand_as(order))
    return rank


def graph_stats(idx, rank):
    """idx (B,N,k) neighbour ids, rank (B,N) position of every point in the order under test."""
    Bq, Nq, k = idx.shape
    r_q = rank[:, :, None].expand(-1, -1, k)
    r_n = torch.gather(rank, 1, idx.reshape(Bq, -1)).reshape(Bq, Nq, k)
    d = (r_q - r_n).abs()
    out = {"near64": float((d <= 64).float().mean()), "near256": float((d <= 256).float().mean())}
    order = rank.argsort(dim=1)                                       # queries walked in the order under test
    nb = torch.gather(r_n, 1, order[:, :, None].expand(-1, -1, k))    # their neighbours, as ranks
    for name, tile in (("targets8", 8), ("targets64", 64)):
        t = nb.reshape(Bq, Nq // tile, tile * k)
        out[name] = float(torch.tensor([[len(torch.unique(row)) for row in blk] for blk in t.cpu()]).float().mean())
    t = (nb // 64).reshape(Bq, Nq // 8, 8 * k)
    out["windows8"] = float(torch.tensor([[len(torch.unique(row)) for row in blk] for blk in t.cpu()]).float().mean())
    return out


from _clouds import structured      # noqa: E402


torch.manual_seed(0)
net = DGCNN([[64, 64]] * 3, [512, 256], 9, k=K, return_edgeconvs=True).to(dev).eval()
for label, x in (("synthetic (bench recipe: uniform xyz, independent colours)", synthetic_blocks(B, N, seed=0)), ("structured (box faces, colour per face)", structured(B, N, 1))):
    x = x.to(dev)
    with torch.no_grad():
        feats = net.edgeconv_features(x)
    ins = [x] + list(feats[:-1])
    ident = torch.arange(N, device=dev).expand(B, -1)
    mort = morton_rank(x[:, 0:3])
    print(label)
    for li, inp in enumerate(ins):
        idx = ops.knn_topk(inp, K)
        a, m = graph_stats(idx, ident), graph_stats(idx, mort)
        print(f"  layer {li + 1} (C = {inp.shape[1]}): given order {a}")
        print(f"  {'':22}Morton order {m}")

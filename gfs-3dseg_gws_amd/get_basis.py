"""Geometric-word extraction with the reference's surface (get_basis.py of the
reference): ``Kmean2Proto``, ``compute_svd``, ``DGCNNSeg`` and a batched,
device-side ``extract_geometric_words`` that does what
``Get_GlobalProto_GlobalKmeans`` (get_basis.py:112-222) does - EdgeConv features
of all base-class points, <=300k points per class, k-means with num_cnt
centroids, per-cluster means, rank-truncated SVD reconstruction - with the
feature extraction on the HIP backbone (batched instead of B=1) and Lloyd's
iterations on the GPU (gfs3d.ops.kmeans_assign) instead of sklearn on the CPU.
"""
import numpy as np
import torch
import torch.nn as nn

from gfs3d import ops
from model.dgcnn import DGCNN


def Kmean2Proto(cls_feat, kmean_label, num_cnt):
    """(N,D) features + (N,) cluster labels -> (num_cnt, D) per-cluster means."""
    rows = []
    for i in range(num_cnt):
        member = kmean_label == i
        assert np.sum(member) != 0
        rows.append(np.mean(cls_feat[member], axis=0))
    return np.stack(rows, axis=0)


def compute_svd(proto_list, keep=0.95):
    """Rank-r reconstruction of the (n,d) prototype matrix, r = smallest rank whose
    singular values sum to more than `keep` of the total (reference get_basis.py:50-71)."""
    u, s, vh = np.linalg.svd(proto_list.transpose(1, 0), full_matrices=False)
    mass = np.cumsum(s)
    above = np.nonzero(mass > keep * np.sum(s))[0]
    r = (int(above[0]) if above.size else len(s) - 1) + 1
    return (u[:, :r] @ np.diag(s[:r]) @ vh[:r, :]).transpose(1, 0)


class DGCNNSeg(nn.Module):
    """Backbone + global-max feature + point-wise segmenter (reference get_basis.py:77-106);
    ``return_feat`` additionally returns the three EdgeConv outputs concatenated (B,192,N)."""

    def __init__(self, args, num_classes):
        super().__init__()
        self.encoder = DGCNN(args.edgeconv_widths, args.dgcnn_mlp_widths, args.pc_in_dim, k=args.dgcnn_k,
                             return_edgeconvs=True)
        in_dim = args.dgcnn_mlp_widths[-1] + sum(w[-1] for w in args.edgeconv_widths)
        self.segmenter = nn.Sequential(
            nn.Conv1d(in_dim, 256, 1, bias=False), nn.BatchNorm1d(256), nn.LeakyReLU(0.2),
            nn.Conv1d(256, 128, 1), nn.BatchNorm1d(128), nn.LeakyReLU(0.2),
            nn.Dropout(0.3), nn.Conv1d(128, num_classes, 1))

    def forward(self, pc, return_feat=False):
        n = pc.shape[2]
        feats, point_feat = self.encoder(pc)
        glob = point_feat.max(dim=-1, keepdim=True)[0].expand(-1, -1, n)
        logits = self.segmenter(torch.cat(feats + [glob], dim=1))
        if return_feat:
            return logits, torch.cat(feats, dim=1)
        return logits


def kmeans_plus_plus(points, n_clusters, rng):
    """k-means++ seeding (host RNG; D^2 sampling evaluated on the device)."""
    M = points.shape[0]
    first = int(rng.randint(M))
    centers = [points[first]]
    d2 = ((points - centers[0]) ** 2).sum(1)
    for _ in range(1, n_clusters):
        prob = (d2 / d2.sum()).double().cpu().numpy()
        nxt = int(rng.choice(M, p=prob / prob.sum()))
        centers.append(points[nxt])
        d2 = torch.minimum(d2, ((points - centers[-1]) ** 2).sum(1))
    return torch.stack(centers, 0)


def device_kmeans(points, n_clusters, init=None, max_iter=300, tol=1e-4, seed=0):
    """Lloyd's algorithm on the GPU.  points (M,D) float32 device tensor.
    Returns (centroids (K,D), labels (M,) int64, inertia float, n_iter)."""
    rng = np.random.RandomState(seed)
    cent = kmeans_plus_plus(points, n_clusters, rng) if init is None else torch.as_tensor(init).to(points)
    cent = cent.contiguous().float()
    scale = float(points.var(dim=0).mean()) * tol
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        _, sums, counts, _ = ops.kmeans_assign(points, cent)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            # points sharded over ranks: one all-reduce of (K x D sums, K counts) per Lloyd iteration
            torch.distributed.all_reduce(sums)
            torch.distributed.all_reduce(counts)
        new = torch.where(counts[:, None] > 0, sums / counts.clamp_min(1)[:, None], cent)
        shift = float(((new - cent) ** 2).sum())
        cent = new
        if shift <= scale:
            break
    label, _, _, inertia = ops.kmeans_assign(points, cent)
    return cent, label.long(), float(inertia.item()), n_iter


@torch.no_grad()
def extract_geometric_words(model, blocks, labels, num_classes, num_cnt, max_per_class=300000, batch=32, seed=0,
                            init=None, max_iter=300):
    """blocks (M,9,N) float, labels (M,N) int (0 = background).  Returns the (num_cnt,192)
    SVD-reconstructed geometric words (numpy), like the pickle the reference writes."""
    model.eval()
    dev = next(model.parameters()).device
    per_class = {c: [] for c in range(1, num_classes)}
    for s in range(0, blocks.shape[0], batch):
        x = blocks[s:s + batch].to(dev)
        y = labels[s:s + batch].to(dev)
        _, feat = model(x, return_feat=True)                  # (b,192,N)
        feat = feat.permute(0, 2, 1).reshape(-1, feat.shape[1])
        y = y.reshape(-1)
        for c in per_class:
            sel = y == c
            if bool(sel.any()):
                per_class[c].append(feat[sel])
    rng = np.random.RandomState(seed)
    chunks = []
    for c in sorted(per_class):
        if not per_class[c]:
            continue
        f = torch.cat(per_class[c], 0)
        if f.shape[0] > max_per_class:
            pick = rng.choice(np.arange(f.shape[0]), max_per_class, replace=False)
            f = f[torch.as_tensor(pick, device=f.device)]
        chunks.append(f)
    pts = torch.cat(chunks, 0).contiguous()
    _, label, _, _ = device_kmeans(pts, num_cnt, init=init, max_iter=max_iter, seed=seed)
    proto = Kmean2Proto(pts.cpu().numpy(), label.cpu().numpy(), num_cnt)
    return compute_svd(proto)

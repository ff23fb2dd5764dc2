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
        from model.dgcnn import segmenter_fusable, segmenter_forward_pm
        if pc.is_cuda and segmenter_fusable(self.segmenter) and self.encoder.fully_fused():
            # point-major throughout: the global max is taken over the points of the (B, N, C) level-2 feature and enters
            # the packed classifier input as a broadcast channel-first source
            feats, feats_pm, level2_pm = self.encoder(pc, point_major=True)
            glob = level2_pm.max(dim=1, keepdim=True)[0].transpose(1, 2).expand(-1, -1, n)
            packed = ops.pack_rows(list(feats_pm) + [glob], [False] * len(feats_pm) + [True])
            logits = segmenter_forward_pm(self.segmenter, packed, self.training, self.encoder.bn_sync)
            if return_feat:
                return logits, torch.cat(feats, dim=1)
            return logits
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


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def _lloyd(points, cent, max_iter, scale, assign_fn, sharded):
    """Lloyd iterations from `cent`.  With `sharded` points every quantity a decision depends on is global: the
    (K x D sums, K counts) are all-reduced each iteration, so `new`, `shift` and hence the stopping iteration are
    identical on all ranks (no rank leaves the loop while another still waits in a collective); an empty cluster
    keeps its previous centroid, which is rank-identical too because the start was."""
    n_iter = 0
    # inside the loop only labels / sums / counts matter: the screened assignment (same labels at about half the matrix work);
    # the final assignment below is the exact form, whose inertia is the one reported
    loop_fn = (lambda p, c: ops.kmeans_assign(p, c, fast=True)) if assign_fn is ops.kmeans_assign else assign_fn
    for n_iter in range(1, max_iter + 1):
        _, sums, counts, _ = loop_fn(points, cent)
        if sharded:
            packed = torch.cat([sums, counts[:, None]], dim=1)          # one collective per iteration
            torch.distributed.all_reduce(packed)
            sums, counts = packed[:, :-1], packed[:, -1]
        new = torch.where(counts[:, None] > 0, sums / counts.clamp_min(1)[:, None], cent)
        shift = float(((new - cent) ** 2).sum())
        cent = new.contiguous()
        if shift <= scale:
            break
    label, _, _, inertia = assign_fn(points, cent)
    if sharded:
        inertia = inertia.clone()
        torch.distributed.all_reduce(inertia)
    return cent, label, float(inertia.item()), n_iter


def device_kmeans(points, n_clusters, init=None, max_iter=300, tol=1e-4, seed=0, n_init=1, assign_fn=None):
    """Lloyd's algorithm on the GPU.  points (M,D) float32 device tensor - under torch.distributed THIS RANK'S SHARD of
    the points.  Returns (centroids (K,D), labels (M,) int64 of the local points, inertia float (global), n_iter).

    `n_init` > 1: that many k-means++ restarts, best inertia wins (the reference's sklearn KMeans ran 10 by default in
    its era, get_basis.py:210; ignored when `init` is given).  Sharded points: the start centroids come from rank 0
    (k-means++ on rank 0's shard, then broadcast), the stop threshold from the all-reduced variance, so every rank runs
    the same number of iterations.  `assign_fn` replaces gfs3d.ops.kmeans_assign (CPU logic tests only)."""
    assign_fn = assign_fn or ops.kmeans_assign
    sharded = _dist_on()
    rng = np.random.RandomState(seed)
    if sharded:                      # global per-feature variance from (count, sum, sum of squares)
        st = torch.stack([torch.full_like(points[0], float(points.shape[0])).double(), points.double().sum(0),
                          (points.double() ** 2).sum(0)])
        torch.distributed.all_reduce(st)
        n_glob, mean = st[0], st[1] / st[0]
        var = ((st[2] / n_glob - mean ** 2) * n_glob / (n_glob - 1).clamp_min(1)).mean()
    else:
        var = points.var(dim=0).mean()
    scale = float(var) * tol
    best = None
    for _ in range(1 if init is not None else max(1, int(n_init))):
        if init is not None:
            cent = torch.as_tensor(init).to(points)
        elif not sharded or torch.distributed.get_rank() == 0:
            cent = kmeans_plus_plus(points, n_clusters, rng)
        else:
            cent = torch.empty((n_clusters, points.shape[1]), dtype=points.dtype, device=points.device)
        cent = cent.contiguous().float()
        if sharded:
            torch.distributed.broadcast(cent, src=0)
        res = _lloyd(points, cent, max_iter, scale, assign_fn, sharded)
        if best is None or res[2] < best[2]:
            best = res
    cent, label, inertia, n_iter = best
    return cent, label.long(), inertia, n_iter


@torch.no_grad()
def extract_geometric_words(model, blocks, labels, num_classes, num_cnt, max_per_class=300000, batch=32, seed=0,
                            init=None, max_iter=300, n_init=1):
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
    _, label, _, _ = device_kmeans(pts, num_cnt, init=init, max_iter=max_iter, seed=seed, n_init=n_init)
    proto = Kmean2Proto(pts.cpu().numpy(), label.cpu().numpy(), num_cnt)
    return compute_svd(proto)

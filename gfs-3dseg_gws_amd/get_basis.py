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


def kmeans_plus_plus_host(points, n_clusters, rng):
    """k-means++ seeding with the D^2 sampling drawn on the host (numpy `rng.choice(M, p=...)`): one device -> host copy of the
    M probabilities per pick.  The form of rounds 1-3; kept as the restatement the device form is tested against and for
    CPU tensors (the gloo logic tests)."""
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


def kmeans_plus_plus(points, n_clusters, rng, return_picks=False):
    """k-means++ seeding (the `init='k-means++'` of the reference's KMeans call, get_basis.py:210).  Device tensors: the whole
    loop runs on the device (gfs3d_kmeanspp_seed_f32) - the host draws the first index and the K - 1 uniform variates from
    `rng` up front, exactly the variates `rng.choice(M, p=...)` would have consumed one per pick, so the picks are those of
    `kmeans_plus_plus_host` (up to a pick whose variate lands within rounding of a cdf step)."""
    if not points.is_cuda:
        cent = kmeans_plus_plus_host(points, n_clusters, rng)
        return (cent, None) if return_picks else cent
    M = points.shape[0]
    first = int(rng.randint(M))
    u = rng.random_sample(n_clusters - 1)
    cent, picks = ops.kmeanspp_seed(points, n_clusters, first, u)
    return (cent, picks) if return_picks else cent


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
    label, sums, counts, inertia = assign_fn(points, cent)
    if sharded:
        K, D = sums.shape
        packed = torch.zeros((K, D + 2), dtype=torch.float64, device=sums.device)     # sums | counts | the inertia (row 0): one collective
        packed[:, :D], packed[:, D], packed[0, D + 1] = sums, counts, inertia.reshape(())
        torch.distributed.all_reduce(packed)
        sums, counts, inertia = packed[:, :D].float(), packed[:, D].float(), packed[0, D + 1].reshape(1)
    return cent, label, float(inertia.item()), n_iter, sums, counts


def device_kmeans(points, n_clusters, init=None, max_iter=300, tol=1e-4, seed=0, n_init=1, assign_fn=None, return_means=False,
                  timings=None):
    """Lloyd's algorithm on the GPU.  points (M,D) float32 device tensor - under torch.distributed THIS RANK'S SHARD of
    the points.  Returns (centroids (K,D), labels (M,) int64 of the local points, inertia float (global), n_iter).

    `n_init` > 1: that many k-means++ restarts, best inertia wins (the reference's sklearn KMeans ran 10 by default in
    its era, get_basis.py:210; ignored when `init` is given).  return_means: also the per-cluster MEANS of the final assignment
    and its member counts (what Kmean2Proto computes on the host, get_basis.py:27-44, from the final assignment's own
    segment sums) - (centroids, labels, inertia, n_iter, means (K,D), counts (K,)).  `timings`: dict receiving the seconds
    spent seeding and in Lloyd's loop.  Sharded points: the start centroids come from rank 0
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
    import time
    best = None
    for _ in range(1 if init is not None else max(1, int(n_init))):
        t0 = time.perf_counter()
        if init is not None:
            cent = torch.as_tensor(init).to(points)
        elif not sharded or torch.distributed.get_rank() == 0:
            cent = kmeans_plus_plus(points, n_clusters, rng)
        else:
            cent = torch.empty((n_clusters, points.shape[1]), dtype=points.dtype, device=points.device)
        cent = cent.contiguous().float()
        if sharded:
            torch.distributed.broadcast(cent, src=0)
        if timings is not None:
            if points.is_cuda:
                torch.cuda.synchronize()
            timings["seeding"] = timings.get("seeding", 0.0) + time.perf_counter() - t0
            t0 = time.perf_counter()
        res = _lloyd(points, cent, max_iter, scale, assign_fn, sharded)
        if timings is not None:
            if points.is_cuda:
                torch.cuda.synchronize()
            timings["lloyd"] = timings.get("lloyd", 0.0) + time.perf_counter() - t0
        if best is None or res[2] < best[2]:
            best = res
    cent, label, inertia, n_iter, sums, counts = best
    if return_means:
        return cent, label.long(), inertia, n_iter, sums / counts.clamp_min(1)[:, None], counts
    return cent, label.long(), inertia, n_iter


@torch.no_grad()
def collect_edgeconv_features(model, blocks, batch=32):
    """EdgeConv features of every point of `blocks` (Mb,9,N), point-major and RESIDENT in HBM: (Mb*N, 192) fp32 (11 GB for
    a 7000-block base split: sized for the card, not for a host round trip per block).  What the reference's B = 1 loop
    collects with `model(ptclouds, return_feat=True)` (get_basis.py:164-170): only the three EdgeConv blocks run - the
    point-wise stack and the segmenter do not feed `feat` - and their point-major outputs land in place, no transposes."""
    enc = model.encoder if hasattr(model, "encoder") else model
    dev = next(enc.parameters()).device
    Mb, _, N = blocks.shape
    width = sum(blk.layer_dims[-1] for blk in enc.edge_convs)
    out = torch.empty((Mb * N, width), dtype=torch.float32, device=dev)
    for s in range(0, Mb, batch):
        x = blocks[s:s + batch].to(dev, non_blocking=True)
        b = x.shape[0]
        feats, feats_pm = enc.edgeconv_features(x, with_point_major=True)
        dst = out[s * N:(s + b) * N].view(b, N, width)
        if all(t is not None for t in feats_pm):
            torch.cat(feats_pm, dim=2, out=dst)
        else:
            dst.copy_(torch.cat(feats, dim=1).permute(0, 2, 1))
    return out


@torch.no_grad()
def subsample_per_class(feat, labels, num_classes, max_per_class=300000, seed=0, sampler="device"):
    """<= max_per_class feature rows per foreground class (reference get_basis.py:187-205: `np.random.choice(..., replace=False)`
    per class, classes concatenated in order).  ONE host synchronisation (the class counts); the class members come from a
    stable sort of the labels (ascending point order inside a class, as the reference's block-order concatenation gives).
    sampler "device": `torch.randperm` on the device from a seeded generator; "numpy": the host draw
    `RandomState(seed).choice(arange(n), max, replace=False)` of rounds 1-3 (a full host permutation per class)."""
    y = labels.reshape(-1).to(feat.device)
    counts = torch.bincount(y, minlength=num_classes)[:num_classes].tolist()           # the one sync
    order = torch.argsort(y, stable=True)
    starts = np.concatenate([[0], np.cumsum(counts)])
    gen = torch.Generator(device=feat.device).manual_seed(seed) if sampler == "device" else None
    rng = np.random.RandomState(seed)
    picks = []
    for c in range(1, num_classes):
        n = counts[c]
        if n == 0:
            continue
        members = order[starts[c]:starts[c] + n]
        if n > max_per_class:
            if sampler == "device":
                members = members[torch.randperm(n, generator=gen, device=feat.device)[:max_per_class]]
            else:
                members = members[torch.as_tensor(rng.choice(np.arange(n), max_per_class, replace=False), device=feat.device)]
        picks.append(members)
    return feat[torch.cat(picks)] if picks else feat[:0]


@torch.no_grad()
def extract_geometric_words(model, blocks, labels, num_classes, num_cnt, max_per_class=300000, batch=32, seed=0,
                            init=None, max_iter=300, n_init=1, sampler="device", timings=None):
    """blocks (M,9,N) float, labels (M,N) int (0 = background).  Returns the (num_cnt,192)
    SVD-reconstructed geometric words (numpy), like the pickle the reference writes (get_basis.py:162-222).
    Device-resident end to end: features -> per-class subsample -> k-means++ -> Lloyd -> per-cluster means (the final
    assignment's own sums / counts, not a host pass over the points) -> 150 KB to the host for the SVD.
    `timings`: a dict that receives the wall-clock seconds of each stage (each closed by a device synchronisation)."""
    import time
    model.eval()

    def lap(name, t0):
        if timings is not None:
            torch.cuda.synchronize()
            timings[name] = timings.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    t = time.perf_counter()
    feat = collect_edgeconv_features(model, blocks, batch)
    t = lap("features", t)
    pts = subsample_per_class(feat, labels, num_classes, max_per_class, seed, sampler).contiguous()
    del feat
    t = lap("subsample", t)
    cent, label, _, n_iter, means, counts = device_kmeans(pts, num_cnt, init=init, max_iter=max_iter, seed=seed, n_init=n_init,
                                                          return_means=True, timings=timings)
    t = time.perf_counter()
    counts_h = counts.cpu().numpy()
    assert (counts_h != 0).all(), "an empty cluster (Kmean2Proto asserts the same, get_basis.py:36)"
    proto = means.cpu().numpy()
    t = lap("means_to_host", t)
    words = compute_svd(proto)
    if timings is not None:
        timings["svd_host"] = time.perf_counter() - t
        timings["lloyd_iterations"] = n_iter
        timings["points"] = int(pts.shape[0])
    return words

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
    """PLAIN k-means++ seeding (one D^2 draw per pick, numpy `rng.choice(M, p=...)` on the host: one device -> host copy of the
    M probabilities per pick).  The form of rounds 1-3; kept as the restatement `gfs3d_kmeanspp_seed_f32` is tested against.
    Not what sklearn runs - see `kmeans_plus_plus`."""
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


def sklearn_local_trials(n_clusters):
    """`n_local_trials = 2 + int(np.log(n_clusters))` (sklearn/cluster/_kmeans.py `_kmeans_plusplus`)."""
    return 2 + int(np.log(n_clusters))


def _draw_seeding_variates(M, n_clusters, n_trials, rng):
    """The variates sklearn's `_kmeans_plusplus` consumes from `rng`, in its order: the first centre through
    `choice(n_samples, p=sample_weight / sample_weight.sum())` with unit float32 weights (one variate), then
    `uniform(size=n_local_trials)` per pick."""
    w = np.ones(M, dtype=np.float32)
    first = int(rng.choice(M, p=w / w.sum()))
    return first, rng.uniform(size=(n_clusters - 1, n_trials))


def kmeans_plus_plus_greedy_host(points, n_clusters, rng, n_local_trials=None):
    """sklearn's greedy k-means++ in tensor ops on whatever device `points` lives on (float64 potentials).  For CPU tensors
    (the gloo logic tests); HIP tensors go through gfs3d_kmeanspp_greedy_f32."""
    M = points.shape[0]
    T = sklearn_local_trials(n_clusters) if n_local_trials is None else int(n_local_trials)
    first, u = _draw_seeding_variates(M, n_clusters, T, rng)
    picks = [first]
    d2 = ((points - points[first]) ** 2).sum(1)
    for t in range(1, n_clusters):
        cs = torch.cumsum(d2.double(), 0)
        r = torch.as_tensor(u[t - 1], dtype=torch.float64, device=points.device) * cs[-1]
        cand = torch.searchsorted(cs, r).clamp_(max=M - 1)
        cand = torch.where(r > 0, cand, torch.zeros_like(cand))
        dc = torch.minimum(((points[None, :, :] - points[cand][:, None, :]) ** 2).sum(2), d2[None, :])
        best = int(torch.argmin(dc.double().sum(1)))
        d2 = dc[best]
        picks.append(int(cand[best]))
    picks = torch.as_tensor(picks, dtype=torch.int64, device=points.device)
    return points[picks], picks


def kmeans_plus_plus(points, n_clusters, rng, return_picks=False, n_local_trials=None):
    """k-means++ seeding as `KMeans(n_clusters, init='k-means++')` of the reference runs it (get_basis.py:210): sklearn's GREEDY
    form - `n_local_trials` = 2 + int(log K) D^2-sampled candidates per pick, the one that lowers the potential most is kept
    (sklearn/cluster/_kmeans.py `_kmeans_plusplus`).  The host draws from `rng` exactly the variates sklearn would, in its
    order, up front; HIP tensors run the whole loop on the device (gfs3d_kmeanspp_greedy_f32), so the picks are sklearn's for
    `random_state=rng` up to rounding-level ties: sklearn's float32 potential sits ~1e-7 of the potential away from an exact sum.
    Measured (tests/golden/kmeanspp*.npz, basis_flow.npz): all picks equal up to 50 000 points, 3 of 4 seeds at 300 000; at the
    reference's full size (2.1e6 points, mean cdf step 5e-7 of the potential) the sequences usually part somewhere - the same
    algorithm on the same variates (statistically equivalent), not seed-for-seed.
    n_local_trials=0: the plain one-draw-per-pick form of rounds 1-4 (`gfs3d_kmeanspp_seed_f32`)."""
    M = points.shape[0]
    if n_local_trials == 0:
        if not points.is_cuda:
            cent = kmeans_plus_plus_host(points, n_clusters, rng)
            return (cent, None) if return_picks else cent
        first = int(rng.randint(M))
        u = rng.random_sample(n_clusters - 1)
        cent, picks = ops.kmeanspp_seed(points, n_clusters, first, u)
        return (cent, picks) if return_picks else cent
    T = sklearn_local_trials(n_clusters) if n_local_trials is None else int(n_local_trials)
    if not points.is_cuda:
        cent, picks = kmeans_plus_plus_greedy_host(points, n_clusters, rng, T)
        return (cent, picks) if return_picks else cent
    if not ops.kmeanspp_greedy_supported(points, T):
        # outside the fused kernel's shapes (C % 4, C <= 256, at most 8 candidates per pick, i.e. K < 1097): the same algorithm
        # in tensor ops on the device - slower, same picks up to the same rounding ties
        cent, picks = kmeans_plus_plus_greedy_host(points, n_clusters, rng, T)
        return (cent, picks) if return_picks else cent
    first, u = _draw_seeding_variates(M, n_clusters, T, rng)
    cent, picks = ops.kmeanspp_greedy(points, n_clusters, first, u, T)
    return (cent, picks) if return_picks else cent


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


def _point_to_centre_sq(points, cent, label, chunk=1 << 18):
    out = torch.empty(points.shape[0], dtype=torch.float32, device=points.device)
    for s in range(0, points.shape[0], chunk):
        out[s:s + chunk] = ((points[s:s + chunk] - cent[label[s:s + chunk].long()]) ** 2).sum(1)
    return out


def _relocate_empty_clusters(points, cent_old, label, sums, counts, sharded):
    """sklearn's `_relocate_empty_clusters_dense` (sklearn/cluster/_k_means_common.pyx; `KMeans.fit` of get_basis.py:210 runs it
    inside every Lloyd iteration): each empty cluster receives one of the n_empty points FARTHEST from their own centres
    (largest first, empty clusters in ascending order); the point leaves its old cluster's sum and count.  `sums` / `counts`
    are the GLOBAL ones under sharding; the candidates are gathered from all ranks so every rank applies the same moves.
    A rare path (a seeding from data points leaves no cluster empty at first): tensor ops, one host read."""
    empty = torch.nonzero(counts == 0).reshape(-1)
    n_empty = int(empty.numel())
    dist = _point_to_centre_sq(points, cent_old, label)
    k = min(n_empty, points.shape[0])
    top = torch.topk(dist, k)
    C = points.shape[1]
    rec = torch.full((n_empty, C + 2), -1.0, dtype=torch.float32, device=points.device)      # distance | old cluster | row
    rec[:k, 0], rec[:k, 1], rec[:k, 2:] = top.values, label[top.indices].float(), points[top.indices]
    if sharded:
        ws = torch.distributed.get_world_size()
        allrec = [torch.empty_like(rec) for _ in range(ws)]
        torch.distributed.all_gather(allrec, rec)
        rec = torch.cat(allrec, 0)
        rec = rec[torch.topk(rec[:, 0], n_empty).indices]          # rank-major order breaks exact ties identically everywhere
    sums, counts = sums.clone(), counts.clone()
    for e in range(n_empty):
        if float(rec[e, 0]) < 0:
            break
        old, row = int(rec[e, 1]), rec[e, 2:]
        sums[old] -= row
        counts[old] -= 1
        sums[int(empty[e])] = row
        counts[int(empty[e])] = 1
    return sums, counts


_DELTA_SHARE = 0.03      # moved points per iteration below which the sums follow the labels (above it a full pass is cheaper)
_RUN_AHEAD = True        # tests switch it off to compare the run-ahead loop with the blocking one


def _spin_until(event):
    """Wait for `event` by POLLING it.  Event.synchronize() may put the host thread to sleep until the completion interrupt; on an
    otherwise idle host the wake-up (a core leaving a deep idle state) cost more than the ~0.1 ms the host has to queue the next
    iteration: measured in the first process on a fresh box, 300 iterations at 1.14 ms instead of 0.63 (scripts/c3_thrice.py)."""
    while not event.query():
        pass


def _lloyd_run_ahead(points, cent, old_label, delta, first_iter, max_iter, scale):
    """The Lloyd iterations from `first_iter` on, once the centroid sums follow the labels (`delta`, single process): the SAME
    iterations as the loop of `_lloyd`, but the host stays one iteration ahead of its own decisions.  Iteration n + 1 (labels from
    the centres of n, the moved rows, the fused tail: three launches) is queued BEFORE the three scalars of iteration n are read -
    read from pinned memory behind an event, so the device never waits for the host (the blocking read cost ~0.1 ms of idle device
    per iteration, a tenth of it).  If iteration n turns out to be the last one, the queued iteration is simply abandoned (the
    final assignment recomputes everything from the returned centres); if it relocates empty clusters, the queued move of the
    running sums is undone - 64-bit integer sums: exactly - and iteration n + 1 is queued again from the relocated centres.
    Returns (centres, n_iter) as the loop would have left them."""
    hosts = [torch.empty(3, dtype=torch.float32).pin_memory() for _ in range(3)]
    events = [torch.cuda.Event() for _ in range(3)]
    slot = [0]

    def launch(cent_in, prev_label):
        label = ops.kmeans_assign(points, cent_in, fast=True, want_sums=False)[0]
        delta.move(prev_label, label)
        sums, counts, new, state = delta.step(cent_in)
        k = slot[0] % 3
        slot[0] += 1
        hosts[k].copy_(state, non_blocking=True)
        events[k].record()
        return dict(label=label, cent_in=cent_in, sums=sums, counts=counts, new=new, k=k)

    n = first_iter
    cur = launch(cent, old_label)
    while True:
        nxt = launch(cur["new"], cur["label"]) if n < max_iter else None       # queued before iteration n is judged
        _spin_until(events[cur["k"]])
        shift, n_changed, n_empty = hosts[cur["k"]].tolist()
        new = cur["new"]
        if n_empty > 0:
            if nxt is not None:
                delta.move(nxt["label"], cur["label"])                          # undo the queued move of the running sums
                nxt = None
            sums, counts = _relocate_empty_clusters(points, cur["cent_in"], cur["label"], cur["sums"], cur["counts"], False)
            new = torch.where(counts[:, None] > 0, sums / counts.clamp_min(1)[:, None], cur["cent_in"]).contiguous()
            shift = float(((new - cur["cent_in"]) ** 2).sum())
        if n_changed == 0 or shift <= scale or n >= max_iter:
            return new, n
        cur = nxt if nxt is not None else launch(new, cur["label"])
        n += 1


def _lloyd(points, cent, max_iter, scale, assign_fn, sharded):
    """Lloyd iterations from `cent` as sklearn's `_kmeans_single_lloyd` runs them (`KMeans.fit`, get_basis.py:210): labels from
    the current centres, member means with empty clusters relocated, stop when the labels repeat (strict convergence) or the
    squared centre shift is <= `scale`; then one exact assignment (labels, sums, counts, inertia of the final centres).
    ONE host read per iteration: (shift, labels changed, empty clusters).  With `sharded` points every quantity a decision
    depends on is global - (K x D sums, K counts, the changed-label count) go through one all-reduce per iteration - so the
    stopping iteration is identical on all ranks (no rank leaves the loop while another still waits in a collective)."""
    n_iter = 0
    # inside the loop only labels / sums / counts matter: the screened assignment (same labels at about half the matrix work);
    # the final assignment below is the exact form, whose inertia is the one reported
    loop_fn = (lambda p, c: ops.kmeans_assign(p, c, fast=True)) if assign_fn is ops.kmeans_assign else assign_fn
    K = cent.shape[0]
    # once few points move per iteration the centroid sums FOLLOW the labels (ops.KMeansDelta: only the moved rows are read,
    # order-independent fixed-point sums) instead of being rebuilt from all the points
    delta_ok = assign_fn is ops.kmeans_assign and points.is_cuda and points.shape[1] % 32 == 0 and 33 <= K <= 256
    delta = None
    m_glob = float(points.shape[0])
    if sharded:
        t = torch.tensor([m_glob], dtype=torch.float64, device=points.device)
        torch.distributed.all_reduce(t)
        m_glob = float(t.item())
    old_label, moved_share = None, 1.0
    for n_iter in range(1, max_iter + 1):
        fused = None
        if delta is not None and old_label is not None and moved_share <= _DELTA_SHARE:
            if not sharded and _RUN_AHEAD:      # the rest of the loop with the host one iteration ahead of its decisions
                cent, n_iter = _lloyd_run_ahead(points, cent, old_label, delta, n_iter, max_iter, scale)
                break
            label = ops.kmeans_assign(points, cent, fast=True, want_sums=False)[0]
            if sharded:
                changed = delta.move(old_label, label).to(torch.float32)
                sums, counts = delta.sums_counts()
            else:                        # one launch: sums, counts, the new centres and the iteration's three scalars
                delta.move(old_label, label)
                sums, counts, new, fused = delta.step(cent)
        else:
            label, sums, counts, _ = loop_fn(points, cent)
            changed = (label != old_label).sum().float().reshape(1) if old_label is not None else \
                torch.full((1,), float(points.shape[0]), device=points.device)
            if delta_ok:
                delta = delta or ops.KMeansDelta(points, K)
                delta.load(sums, counts)
        if sharded:
            D = sums.shape[1]
            packed = torch.zeros((K + 1, D + 1), dtype=sums.dtype, device=sums.device)   # sums | counts ; changed: one collective
            packed[:K, :D], packed[:K, D], packed[K, 0] = sums, counts, changed[0]
            torch.distributed.all_reduce(packed)
            sums, counts, changed = packed[:K, :D], packed[:K, D], packed[K, :1]
        if fused is not None:
            state = fused
        else:
            new = torch.where(counts[:, None] > 0, sums / counts.clamp_min(1)[:, None], cent)
            state = torch.cat([((new - cent) ** 2).sum().reshape(1), changed.to(new.dtype), (counts == 0).sum().reshape(1).to(new.dtype)])
        shift, n_changed, n_empty = state.tolist()                                      # the iteration's one host read
        moved_share = n_changed / m_glob
        if n_empty > 0:
            sums, counts = _relocate_empty_clusters(points, cent, label, sums, counts, sharded)
            new = torch.where(counts[:, None] > 0, sums / counts.clamp_min(1)[:, None], cent)
            shift = float(((new - cent) ** 2).sum())
        cent = new.contiguous()
        if n_changed == 0 or shift <= scale:
            break
        old_label = label
    label, sums, counts, inertia = assign_fn(points, cent)
    if sharded:
        K, D = sums.shape
        packed = torch.zeros((K, D + 2), dtype=torch.float64, device=sums.device)     # sums | counts | the inertia (row 0): one collective
        packed[:, :D], packed[:, D], packed[0, D + 1] = sums, counts, inertia.reshape(())
        torch.distributed.all_reduce(packed)
        sums, counts, inertia = packed[:, :D].float(), packed[:, D].float(), packed[0, D + 1].reshape(1)
    return cent, label, float(inertia.item()), n_iter, sums, counts


def device_kmeans(points, n_clusters, init=None, max_iter=300, tol=1e-4, seed=0, n_init=1, assign_fn=None, return_means=False,
                  timings=None, rng=None, center=True, n_local_trials=None, details=None):
    """`KMeans(n_clusters, init='k-means++').fit(points)` (get_basis.py:210) on the GPU, step for step as sklearn's `fit`:
    centre the data on its mean, greedy k-means++ from `rng` (default `RandomState(seed)`), Lloyd iterations with empty-cluster
    relocation, stop on repeated labels or a centre shift <= tol * mean(var(points, axis=0)), final assignment, centres shifted
    back.  points (M,D) float32 device tensor - under torch.distributed THIS RANK'S SHARD of the points.  Returns
    (centroids (K,D), labels (M,) int64 of the local points, inertia float (global), n_iter).

    `n_init` > 1: that many k-means++ restarts, best inertia wins (the reference's sklearn KMeans ran 10 by default in
    its era, sklearn >= 1.4 runs 1; ignored when `init` is given).  return_means: also the per-cluster MEANS of the final
    assignment and its member counts (what Kmean2Proto computes on the host, get_basis.py:27-44, from the final assignment's own
    segment sums) - (centroids, labels, inertia, n_iter, means (K,D), counts (K,)).  `timings`: dict receiving the seconds
    spent seeding and in Lloyd's loop.  Sharded points: the start centroids come from rank 0 (k-means++ on rank 0's shard,
    then broadcast), mean and stop threshold from all-reduced moments, so every rank runs the same number of iterations.
    `assign_fn` replaces gfs3d.ops.kmeans_assign (CPU logic tests only).  center=False: iterate on the data as given;
    center="inplace": the caller gives `points` up - they are centred in place (no second 1.6 GB copy at configs[2] size; the
    labels, means and centres returned are the same, the caller's tensor is left centred).  `details`: dict receiving the
    seeding's `picks` (indices into `points`; rank 0's shard under sharding)."""
    assign_fn = assign_fn or ops.kmeans_assign
    sharded = _dist_on()
    rng = rng if rng is not None else np.random.RandomState(seed)
    if sharded:                      # global per-feature mean and (biased, as numpy's) variance from (count, sum, sum of squares)
        st = torch.stack([torch.full_like(points[0], float(points.shape[0])).double(), points.double().sum(0),
                          (points.double() ** 2).sum(0)])
        torch.distributed.all_reduce(st)
        mean = st[1] / st[0]
        var = (st[2] / st[0] - mean ** 2).mean()
        mean = mean.to(points.dtype)
    else:
        var = points.var(dim=0, unbiased=False).mean()
        mean = points.mean(dim=0)
    scale = float(var) * tol
    if center == "inplace":
        points.sub_(mean)
    elif center:
        points = points - mean
    import time
    best = None
    for _ in range(1 if init is not None else max(1, int(n_init))):
        t0 = time.perf_counter()
        if init is not None:
            cent = torch.as_tensor(init).to(points)
            if center:
                cent = cent - mean
        elif not sharded or torch.distributed.get_rank() == 0:
            cent, picks = kmeans_plus_plus(points, n_clusters, rng, return_picks=True, n_local_trials=n_local_trials)
            if details is not None:
                details["picks"] = picks
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
    means = sums / counts.clamp_min(1)[:, None]
    if center:
        cent, means = cent + mean, means + mean
    if return_means:
        return cent, label.long(), inertia, n_iter, means, counts
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
def subsample_per_class(feat, labels, num_classes, max_per_class=300000, seed=0, sampler="device", rng=None, details=None):
    """<= max_per_class feature rows per foreground class (reference get_basis.py:187-205: `np.random.choice(..., replace=False)`
    per class, classes concatenated in order).  ONE host synchronisation (the class counts); the class members come from a
    stable sort of the labels (ascending point order inside a class, as the reference's block-order concatenation gives).
    sampler "device": `torch.randperm` on the device from a seeded generator; "numpy": the reference's own host draw
    `rng.choice(arange(n), max, replace=False)` per over-full class, in class order, from `rng` (default `RandomState(seed)`) -
    pass the RandomState that `device_kmeans` continues with and the two consume ONE stream, as the reference's
    `np.random.seed(s)` ... `np.random.choice` ... `KMeans(random_state=None)` does (get_basis.py:310, 198, 210).
    `details`: dict receiving `sub_idx` = the per-class draws (class -> indices into the class's rows)."""
    y = labels.reshape(-1).to(feat.device).long()
    # labels outside [0, num_classes) - an ignore index such as -100, float labels - match no `y == c` mask of the reference
    # (get_basis.py:176-181): they are skipped, i.e. counted with the background
    y = torch.where((y < 0) | (y >= num_classes), torch.zeros_like(y), y)
    counts = torch.bincount(y, minlength=num_classes)[:num_classes].tolist()           # the one sync
    order = torch.argsort(y, stable=True)
    starts = np.concatenate([[0], np.cumsum(counts)])
    if sampler not in ("device", "numpy"):
        raise ValueError(f"subsample_per_class: sampler must be 'device' or 'numpy', got {sampler!r}")
    gen = torch.Generator(device=feat.device).manual_seed(seed) if sampler == "device" else None
    rng = rng if rng is not None else np.random.RandomState(seed)
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
                drawn = rng.choice(np.arange(n), max_per_class, replace=False)
                if details is not None:
                    details.setdefault("sub_idx", {})[c] = drawn
                members = members[torch.as_tensor(drawn, device=feat.device)]
        picks.append(members)
    return feat[torch.cat(picks)] if picks else feat[:0]


@torch.no_grad()
def extract_geometric_words(model, blocks, labels, num_classes, num_cnt, max_per_class=300000, batch=32, seed=0,
                            init=None, max_iter=300, n_init=1, sampler="device", timings=None, rng=None, details=None):
    """blocks (M,9,N) float, labels (M,N) int (0 = background).  Returns the (num_cnt,192)
    SVD-reconstructed geometric words (numpy), like the pickle the reference writes (get_basis.py:162-222).
    Device-resident end to end: features -> per-class subsample -> k-means++ -> Lloyd -> per-cluster means (the final
    assignment's own sums / counts, not a host pass over the points) -> 150 KB to the host for the SVD.
    `timings`: a dict that receives the wall-clock seconds of each stage (each closed by a device synchronisation).
    `rng`: ONE numpy RandomState for the whole flow - the per-class subsample (sampler "numpy") and the k-means++ seeding draw
    from it in sequence, which is what the reference does with numpy's global stream after its single `np.random.seed(args.seed)`
    (get_basis.py:310 -> :198 -> :210).  `rng=np.random.RandomState(s), sampler="numpy"` therefore reproduces the reference's
    subsample rows and seeding variates for `--seed s` (tests/golden/basis_flow.npz); default: separate streams from `seed`.
    `details`: dict receiving the subsample draws, the seeding picks, n_iter, inertia, the cluster means."""
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
    pts = subsample_per_class(feat, labels, num_classes, max_per_class, seed, sampler, rng=rng, details=details).contiguous()
    del feat
    t = lap("subsample", t)
    cent, label, inertia, n_iter, means, counts = device_kmeans(pts, num_cnt, init=init, max_iter=max_iter, seed=seed, n_init=n_init,
                                                                return_means=True, timings=timings, rng=rng, details=details,
                                                                center="inplace")
    if details is not None:
        details.update(n_iter=n_iter, inertia=inertia, means=means, counts=counts, rows=int(pts.shape[0]))
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

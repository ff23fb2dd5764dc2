"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain torch-CPU / numpy ops, of the reference hot path
(Pixie8888/GFS-3DSeg_GWs).  Every function cites the reference file:line it
follows.  It is functional (weights come in as a ``state_dict``-style mapping
with the reference's key names) so it shares no module structure with the
product code under ``gfs-3dseg_gws_amd/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker.

Parity pinning: ``oracle/gen_golden.py`` ran the *real* reference (imported
from /root/reference in the build container) on seeded inputs and committed the
results under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
file against those fixtures.
"""
from __future__ import annotations

import math
import random as _pyrandom
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Mapping[str, Tensor]

BN_EPS = 1e-5        # nn.BatchNorm default (model/dgcnn.py:54,73)
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.2    # model/dgcnn.py:56,75


# --------------------------------------------------------------------------
# kNN graph  (model/dgcnn.py:17-23)
# --------------------------------------------------------------------------
def pairwise_neg_sqdist(x: Tensor) -> Tensor:
    """pd[b,i,j] = -|xj|^2 + 2 xi.xj - |xi|^2 with the reference's operand
    order: ``-xx - inner - xx^T`` where inner = -2 x^T x (dgcnn.py:18-20)."""
    inner = -2.0 * torch.matmul(x.transpose(2, 1), x)
    sq = torch.sum(x * x, dim=1, keepdim=True)
    return (-sq - inner) - sq.transpose(2, 1)


def knn(x: Tensor, k: int) -> Tensor:
    """(B,C,N) -> (B,N,k) int64, nearest first (dgcnn.py:22)."""
    return pairwise_neg_sqdist(x).topk(k=k, dim=-1)[1]


def knn_exact_order(x: Tensor, k: int) -> Tensor:
    """The *build's* deterministic kNN contract (SURVEY.md §7 'Hard parts'):
    dot products as a c-ordered fp32 fma chain, same operand order as
    dgcnn.py:18-20 afterwards, ties resolved to the lowest index.  The C file
    ``oracle/knn_oracle.c`` is the fast version of exactly this function; the
    HIP kernel must equal it bit for bit."""
    B, C, N = x.shape
    xf = x.to(torch.float32).numpy()
    out = np.empty((B, N, k), dtype=np.int64)
    for b in range(B):
        dot = np.zeros((N, N), dtype=np.float32)
        for c in range(C):  # fp32 fma chain in c order (fma emulated in fp64: exact product, one rounding)
            dot = (xf[b, c][:, None].astype(np.float64) * xf[b, c][None, :].astype(np.float64)
                   + dot.astype(np.float64)).astype(np.float32)
        sq = np.zeros((N,), dtype=np.float32)
        for c in range(C):
            sq = (sq + (xf[b, c] * xf[b, c]).astype(np.float32)).astype(np.float32)
        pd = ((-sq[None, :] + np.float32(2.0) * dot).astype(np.float32) - sq[:, None]).astype(np.float32)
        # stable descending: lowest index first among equal values
        order = np.argsort(-pd, axis=1, kind="stable")
        out[b] = order[:, :k]
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# edge features (model/dgcnn.py:26-42)
# --------------------------------------------------------------------------
def edge_feature(x: Tensor, k: int, idx: Optional[Tensor] = None) -> Tensor:
    """(B,C,N) -> (B,2C,N,k): cat(x_j - x_i, x_i) on the channel axis."""
    B, C, N = x.shape
    if idx is None:
        idx = knn(x, k)
    flat = idx.reshape(B, 1, N * k).expand(B, C, N * k)
    nbr = torch.gather(x, 2, flat).reshape(B, C, N, k)
    ctr = x.unsqueeze(-1).expand(B, C, N, k)
    return torch.cat((nbr - ctr, ctr), dim=1)


# --------------------------------------------------------------------------
# 1x1 conv + BN + LeakyReLU stacks (model/dgcnn.py:45-80)
# --------------------------------------------------------------------------
def _bn(h: Tensor, p: Params, prefix: str, training: bool,
        new_stats: Optional[Dict[str, Tensor]]) -> Tensor:
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        out = F.batch_norm(h, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
        if new_stats is not None:
            new_stats[prefix + ".running_mean"] = rm2
            new_stats[prefix + ".running_var"] = rv2
        return out
    return F.batch_norm(h, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)


def conv_stack(h: Tensor, p: Params, prefix: str, n_layers: int, training: bool,
               new_stats: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """``conv2d`` / ``conv1d`` blocks: [conv1x1(bias=False) -> BN -> LReLU(0.2)]*n
    stored as ``<prefix>.layer.{3i, 3i+1}`` (dgcnn.py:52-57, 71-76)."""
    for i in range(n_layers):
        w = p[f"{prefix}.layer.{3 * i}.weight"]
        h = F.conv2d(h, w) if w.dim() == 4 else F.conv1d(h, w)
        h = _bn(h, p, f"{prefix}.layer.{3 * i + 1}", training, new_stats)
        h = F.leaky_relu(h, LRELU_SLOPE)
    return h


def dgcnn_forward(p: Params, x: Tensor, k: int = 20, n_edgeconv: int = 3, n_conv_per_block: int = 2,
                  n_mlp: int = 2, training: bool = False, prefix: str = "",
                  new_stats: Optional[Dict[str, Tensor]] = None,
                  knn_fn=knn) -> Tuple[List[Tensor], Tensor]:
    """DGCNN.forward (model/dgcnn.py:113-127): 3x (edge feature -> conv2d block
    -> max over k), kNN recomputed on each layer's features; cat -> conv1d."""
    outs: List[Tensor] = []
    h = x
    for i in range(n_edgeconv):
        e = edge_feature(h, k, knn_fn(h, k))
        e = conv_stack(e, p, f"{prefix}edge_convs.{i}", n_conv_per_block, training, new_stats)
        h = e.max(dim=-1)[0]
        outs.append(h)
    out = conv_stack(torch.cat(outs, dim=1), p, f"{prefix}conv", n_mlp, training, new_stats)
    return outs, out


# --------------------------------------------------------------------------
# BaseLearner (model/dgcnn.py:130-152) and SelfAttention (model/attention.py:39-48)
# --------------------------------------------------------------------------
def base_learner(p: Params, x: Tensor, n_convs: int, training: bool, prefix: str = "base_learner.",
                 new_stats: Optional[Dict[str, Tensor]] = None) -> Tensor:
    for i in range(n_convs):
        x = F.conv1d(x, p[f"{prefix}convs.{i}.0.weight"], p[f"{prefix}convs.{i}.0.bias"])
        x = _bn(x, p, f"{prefix}convs.{i}.1", training, new_stats)
        if i != n_convs - 1:
            x = F.relu(x)
    return x


def self_attention(p: Params, x: Tensor, prefix: str = "att_learner.",
                   dropout_mask: Optional[Tensor] = None) -> Tensor:
    """q/k/v 1x1 conv, softmax(q^T k / sqrt(d)) over the last axis, y = attn v^T
    (attention.py:39-48).  ``dropout_mask`` (already scaled by 1/(1-p)) stands in
    for nn.Dropout(0.1) in train mode; None = eval."""
    q = F.conv1d(x, p[prefix + "q_map.weight"])
    kk = F.conv1d(x, p[prefix + "k_map.weight"])
    v = F.conv1d(x, p[prefix + "v_map.weight"])
    temp = q.shape[1] ** 0.5
    attn = F.softmax(torch.matmul(q.transpose(1, 2) / temp, kk), dim=-1)
    if dropout_mask is not None:
        attn = attn * dropout_mask
    return torch.matmul(attn, v.transpose(1, 2)).transpose(1, 2)


# --------------------------------------------------------------------------
# geometric words / prototypes (model/capl.py)
# --------------------------------------------------------------------------
def geometric_word_assign(gp: Tensor, edge_convs: Tensor) -> Tuple[Tensor, Tensor]:
    """capl.py:344-353: cosine of every point's 192-d EdgeConv feature with every
    geometric word, softmax(10*cos) over words, arg-max one-hot."""
    ec = F.normalize(edge_convs, p=2, dim=1)
    g = F.normalize(gp, p=2, dim=1).unsqueeze(0)
    soft = torch.softmax(10 * torch.matmul(g, ec), dim=1)
    onehot = F.one_hot(torch.argmax(soft, dim=1), num_classes=gp.shape[0]).transpose(2, 1).float()
    return soft, onehot


def get_features(p: Params, gp: Tensor, x: Tensor, k: int = 20, n_base_convs: int = 2,
                 training: bool = False, new_stats: Optional[Dict[str, Tensor]] = None,
                 dropout_mask: Optional[Tensor] = None, knn_fn=knn, encoder_out=None) -> Tuple[Tensor, Tensor, Tensor]:
    """capl.py:324-362.  `encoder_out` = (three EdgeConv outputs, level-2 feature): run only the heads on given
    backbone outputs (tests: gradient parity of the heads on bit-identical inputs)."""
    if encoder_out is not None:
        ecs, lvl2 = encoder_out
    else:
        ecs, lvl2 = dgcnn_forward(p, x, k=k, training=training, prefix="encoder.", new_stats=new_stats, knn_fn=knn_fn)
    lvl3 = base_learner(p, lvl2, n_base_convs, training, new_stats=new_stats)
    att = self_attention(p, lvl2, dropout_mask=dropout_mask)
    semantic = torch.cat((ecs[0], att, lvl3), dim=1)
    soft, onehot = geometric_word_assign(gp, torch.cat(ecs, dim=1))
    fused = F.conv1d(torch.cat([soft, semantic], dim=1), p["fusion.0.weight"], p["fusion.0.bias"])
    fused = _bn(fused, p, "fusion.1", training, new_stats)
    return F.leaky_relu(fused, LRELU_SLOPE), semantic, onehot


def get_fg_feat(p: Params, gp: Tensor, x: Tensor, y: Tensor, k: int = 20, knn_fn=knn) -> Tuple[Tensor, Tensor]:
    """capl.py:71-88: features of ONE support cloud, foreground columns (y == 1) only ->
    (n_fg, 128) fused features and (n_fg, K_w) one-hot words."""
    pf, _, onehot = get_features(p, gp, x, k=k, training=False, knn_fn=knn_fn)
    keep = y[0] == 1
    return pf[0][:, keep].transpose(1, 0), onehot[0][:, keep].transpose(1, 0)


def get_pred(x: Tensor, proto: Tensor, bg_proto: Optional[Tensor] = None) -> Tensor:
    """capl.py:290-322: 10 * cosine(point feature, prototype); optional bg row
    prepended; proto is (cls,c) or per-sample (b,cls,c)."""
    if proto.dim() == 3:
        if bg_proto is not None:
            proto = torch.cat([bg_proto.unsqueeze(0).repeat(proto.shape[0], 1, 1), proto], dim=1)
        pn = F.normalize(proto, p=2, dim=-1)
    else:
        if bg_proto is not None:
            proto = torch.cat([bg_proto, proto], dim=0)
        pn = F.normalize(proto, p=2, dim=1).unsqueeze(0)
    return (pn @ F.normalize(x, p=2, dim=1)) * 10


def post_refine_proto_v2(proto: Tensor, x: Tensor, point_feat: Tensor,
                         bg_proto: Optional[Tensor] = None) -> Tensor:
    """capl.py:245-287 (softmax over the POINT axis, dim=2)."""
    pred = F.softmax(get_pred(x, proto, bg_proto), 2)
    pp = pred @ point_feat.permute(0, 2, 1)
    if bg_proto is not None:
        pp = pp[:, 1:, :]
    w = (F.normalize(pp, 2, -1) * F.normalize(proto, 2, -1).unsqueeze(0)).sum(-1, keepdim=True)
    w = w * (w > 0).to(w.dtype)
    return w * pp + (1 - w) * proto.unsqueeze(0)


def get_gp_weight(coding: Tensor, onehot: Tensor, base_num: int, th: float, gt_label: Optional[Tensor] = None,
                  use_bg_weight: bool = False):
    """capl.py:92-142."""
    score = torch.matmul(coding.unsqueeze(0).repeat(onehot.shape[0], 1, 1), onehot)
    acc, novel_acc = 0.0, 0.0
    if gt_label is not None and not use_bg_weight:
        g = F.one_hot(gt_label, num_classes=score.shape[1]).transpose(2, 1)
        per_pt = torch.sum(g * score, dim=1)
        acc = torch.mean(per_pt)
        nm = gt_label > base_num - 1
        novel_acc = torch.mean(per_pt[nm]) if torch.sum(nm) > 0 else torch.zeros_like(acc)
    elif gt_label is not None and use_bg_weight:
        g = F.one_hot(gt_label, num_classes=score.shape[1] + 1).transpose(2, 1)[:, 1:, :]
        acc = torch.mean(torch.sum(g * score, dim=1))
    weight = torch.ones_like(score)
    weight[score == 1] = th
    if use_bg_weight:
        weight = torch.cat([torch.ones_like(weight[:, :1]), weight], dim=1)
        gm = F.one_hot(gt_label, num_classes=score.shape[1] + 1).transpose(2, 1)
        weight[gm == 1] = 1.0
    return weight, acc, novel_acc


def hard_coding(coding: Tensor, energy: float) -> Tensor:
    """capl.py:413-433 / train.py:132-152: keep the smallest descending prefix
    whose mass exceeds energy*total; multi-hot.  (Does not modify its input.)"""
    order = torch.argsort(coding, descending=True)
    total = torch.sum(coding)
    acc = 0.0
    mask = torch.zeros_like(coding)
    for i in order:
        acc = acc + coding[i]
        mask[i] = 1.0
        if acc > energy * total:
            break
    return mask.clone()


def generate_fake_proto(x: Tensor, y: Tensor, main_proto: Tensor, fake_novel: Optional[Sequence] = None):
    """capl.py:364-411 (without the hard-wired .cuda())."""
    ty = y.unsqueeze(1)
    if fake_novel is None:
        uniq = list(ty.unique())
        if 0 in uniq:
            uniq.remove(0)
        fake_novel = _pyrandom.sample(uniq, len(uniq) // 2)
    proto = main_proto / (torch.norm(main_proto, 2, 1, True) + 1e-12)
    xn = x / (torch.norm(x, 2, 1, True) + 1e-12)
    for fn in fake_novel:
        m = (ty == fn).to(xn.dtype)
        mean = (xn * m).sum(0).sum(-1) / (m.sum(0).sum(-1) + 1e-12)
        sel = torch.zeros(proto.size(0), 1, dtype=proto.dtype, device=proto.device)
        sel[int(fn) - 1] = 1
        proto = proto * (1 - sel) + mean.unsqueeze(0) * sel
    return proto, fake_novel


def capl_forward_eval(p: Params, gp: Tensor, x: Tensor, y: Optional[Tensor], gened_proto: Tensor,
                      base_coding: Tensor, novel_coding: Tensor, base_num: int, eval_weight: float,
                      k: int = 20, knn_fn=knn):
    """capl.py:170-192."""
    pf, _, onehot = get_features(p, gp, x, k=k, training=False, knn_fn=knn_fn)
    if gened_proto.dim() == 3:
        gened_proto = gened_proto[0]
    rp = post_refine_proto_v2(p["main_proto"], pf, pf)
    rp[:, :base_num] = rp[:, :base_num] + gened_proto[:base_num].unsqueeze(0)
    rp[:, base_num:] = rp[:, base_num:] * 0 + gened_proto[base_num:].unsqueeze(0)
    logits = get_pred(pf, rp)
    w, acc, nacc = get_gp_weight(torch.cat([base_coding, novel_coding], 0), onehot, base_num, eval_weight, gt_label=y)
    return logits * w, acc, nacc


def capl_forward_train(p: Params, gp: Tensor, x: Tensor, y: Tensor, base_num: int, k: int = 20,
                       fake_novel: Optional[Sequence] = None, new_stats=None, dropout_mask=None, knn_fn=knn,
                       encoder_out=None):
    """capl.py:194-242: two CE losses on cosine logits with fake-novel prototypes."""
    pf, _, _ = get_features(p, gp, x, k=k, training=True, new_stats=new_stats, dropout_mask=dropout_mask, knn_fn=knn_fn,
                            encoder_out=encoder_out)
    logits2, loss, fake_novel = capl_train_head(pf, y, p["main_proto"], p["bg_proto"], base_num, fake_novel)
    return logits2.max(1)[1], loss, fake_novel


def capl_train_head(pf: Tensor, y: Tensor, main_proto: Tensor, bg_proto: Tensor, base_num: int,
                    fake_novel: Optional[Sequence] = None, ignore_index: int = -100):
    """capl.py:194-242 after getFeatures: fake-novel prototypes, first cosine head + CE, prototype refinement, second cosine
    head + CE (criterion = CrossEntropyLoss(ignore_index), train.py:317).  Returns (logits_2, loss, fake_novel).  Works in
    float32 and float64 (the GPU parity test of the fused head evaluates it in float64)."""
    half = pf.size(0) // 2
    ori, fake_novel = generate_fake_proto(pf[half:], y[half:], main_proto.clone(), fake_novel)
    l1 = F.cross_entropy(get_pred(pf, ori, bg_proto), y, ignore_index=ignore_index)
    rp = post_refine_proto_v2(main_proto.clone(), pf, pf, bg_proto).clone()
    rp[:, :base_num] = rp[:, :base_num] + ori[:base_num].unsqueeze(0)
    rp[:, base_num:] = rp[:, base_num:] * 0 + ori[base_num:].unsqueeze(0)
    logits2 = get_pred(pf, rp, bg_proto)
    l2 = F.cross_entropy(logits2, y, ignore_index=ignore_index)
    return logits2, 0.5 * l2 + 0.5 * l1, fake_novel


# --------------------------------------------------------------------------
# get_basis.py pieces
# --------------------------------------------------------------------------
def kmean2proto(feat: np.ndarray, label: np.ndarray, n: int) -> np.ndarray:
    """get_basis.py:27-44: per-cluster mean."""
    return np.stack([feat[label == i].mean(axis=0) for i in range(n)], axis=0)


def svd_reconstruct(proto: np.ndarray, keep: float = 0.95) -> np.ndarray:
    """get_basis.py:50-71: smallest rank r with sum(s[:r]) > keep*sum(s); rank-r rebuild."""
    u, s, vh = np.linalg.svd(proto.transpose(1, 0), full_matrices=False)
    r = len(s)
    for i in range(len(s)):
        if np.sum(s[: i + 1]) > keep * np.sum(s):
            r = i + 1
            break
    return (u[:, :r] @ np.diag(s[:r]) @ vh[:r, :]).transpose(1, 0)


def lloyd_kmeans(pts: np.ndarray, init: np.ndarray, iters: int) -> Tuple[np.ndarray, np.ndarray, float]:
    """Plain Lloyd iterations from a fixed init (what sklearn KMeans(init=<array>,
    n_init=1, algorithm='lloyd') converges through, get_basis.py:210).  Returns
    (centroids, labels, inertia); empty clusters keep their previous centroid."""
    c = init.astype(np.float64).copy()
    x = pts.astype(np.float64)
    lab = np.zeros(len(x), dtype=np.int64)
    for _ in range(iters):
        d = (x * x).sum(1, keepdims=True) - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
        lab = d.argmin(1)
        for j in range(len(c)):
            m = lab == j
            if m.any():
                c[j] = x[m].mean(0)
    d = (x * x).sum(1, keepdims=True) - 2.0 * x @ c.T + (c * c).sum(1)[None, :]
    lab = d.argmin(1)
    return c.astype(np.float32), lab, float(d[np.arange(len(x)), lab].sum())


# --------------------------------------------------------------------------
# sklearn.cluster.KMeans(n_clusters, init='k-means++') as the reference calls it (get_basis.py:210).
# scikit-learn is a third-party dependency of the reference, not vendored under /root/reference and not pinned by it
# (README names no version); the algorithm restated here is scikit-learn 1.7.2's (the version in the build container, the
# one oracle/gen_golden.py ran): sklearn/cluster/_kmeans.py `_kmeans_plusplus`, `_kmeans_single_lloyd`,
# `_relocate_empty_clusters` (sklearn/cluster/_k_means_common.pyx) and `KMeans.fit`.  Pinned by tests/golden/kmeanspp.npz,
# which holds what sklearn itself returned.
# --------------------------------------------------------------------------
def clustered_points(M: int, C: int, n_blobs: int, seed: int, spread: float = 0.35) -> np.ndarray:
    """Input recipe of the k-means fixtures (legacy numpy RandomState: the same bytes on every platform): `n_blobs` Gaussian
    blobs of unequal size around an offset mean, float32 (M,C) - shaped like EdgeConv features (not centred)."""
    g = np.random.RandomState(seed)
    cent = g.randn(n_blobs, C) * g.uniform(0.2, 1.0, size=(1, C)) + g.uniform(-0.5, 1.5, size=(1, C))
    w = g.uniform(0.2, 1.0, size=n_blobs)
    which = g.choice(n_blobs, size=M, p=w / w.sum())
    return (cent[which] + spread * g.randn(M, C) * g.uniform(0.5, 1.5, size=(n_blobs, 1))[which]).astype(np.float32)


def sklearn_first_center(M: int, rs: np.random.RandomState) -> int:
    """`random_state.choice(n_samples, p=sample_weight / sample_weight.sum())` with unit float32 weights
    (_kmeans_plusplus): consumes ONE uniform variate."""
    w = np.ones(M, dtype=np.float32)
    return int(rs.choice(M, p=w / w.sum()))


def _sqdist_rows(X: np.ndarray, rows: np.ndarray, chunk: int = 8192) -> np.ndarray:
    """(T,M) squared distances of X[rows] to every point, evaluated in float64 and rounded to float32 - what sklearn's
    `_euclidean_distances` returns for float32 input (it upcasts blocks to float64, `_euclidean_distances_upcast`)."""
    Y = X[rows].astype(np.float64)
    out = np.empty((len(rows), X.shape[0]), dtype=np.float32)
    for s in range(0, X.shape[0], chunk):
        x = X[s:s + chunk].astype(np.float64)
        out[:, s:s + chunk] = ((x[None, :, :] - Y[:, None, :]) ** 2).sum(2)
    return out


def greedy_kmeans_plusplus(X: np.ndarray, K: int, rs: np.random.RandomState, n_local_trials: Optional[int] = None,
                           with_margins: bool = False):
    """sklearn's greedy k-means++ (`_kmeans_plusplus`): per pick `2 + int(log K)` candidates drawn by D^2 sampling
    (`searchsorted(cumsum(d2), uniform * potential)`, side 'left', clipped), the one that lowers the potential most is kept.
    The potentials are summed in float64 here (sklearn: a float32 BLAS dot, whose summation order is the library's own).
    Returns (indices (K,), [margins (K,2)]): margins[c] = (smallest distance of a scaled variate to a cdf step, gap between
    the two lowest distinct candidate potentials), both relative to the potential - how close pick c came to a rounding tie."""
    M = X.shape[0]
    T = 2 + int(np.log(K)) if n_local_trials is None else int(n_local_trials)
    idx = np.full(K, -1, dtype=np.int64)
    idx[0] = sklearn_first_center(M, rs)
    d2 = _sqdist_rows(X, idx[:1])[0]
    pot = float(d2.astype(np.float64).sum())
    margins = np.ones((K, 2))
    for c in range(1, K):
        r = rs.uniform(size=T) * pot
        cs = np.cumsum(d2, dtype=np.float64)
        cand = np.searchsorted(cs, r)
        np.clip(cand, None, M - 1, out=cand)
        dc = np.minimum(_sqdist_rows(X, cand), d2[None, :])
        pots = dc.astype(np.float64).sum(1)
        best = int(np.argmin(pots))
        if with_margins and pot > 0:
            lo = np.where(cand > 0, cs[np.maximum(cand - 1, 0)], 0.0)
            step = np.minimum(r - lo, cs[cand] - r).min() / pot
            up = np.unique(pots)
            margins[c] = (step, (up[1] - up[0]) / pot if len(up) > 1 else 1.0)
        pot, d2, idx[c] = float(pots[best]), dc[best], cand[best]
    return (idx, margins) if with_margins else idx


def sklearn_lloyd(X: np.ndarray, init: np.ndarray, max_iter: int = 300, tol: float = 1e-4, chunk: int = 16384):
    """`KMeans.fit` after the seeding: Lloyd iterations as `_kmeans_single_lloyd` runs them.  Per iteration: labels from the
    current centres, new centres = member means with EMPTY clusters relocated to the points farthest from their centres
    (`_relocate_empty_clusters`: the n_empty largest distances, each point leaves its old cluster), stop when the labels
    repeat (strict convergence) or the squared centre shift is <= tol * mean(var(X, axis=0)); a last assignment when the
    stop was the shift.  float64 arithmetic (sklearn: float32 GEMM chunks).  Returns (centres, labels, inertia, n_iter)."""
    x = X.astype(np.float64)
    c = np.asarray(init, dtype=np.float64).copy()
    K = len(c)
    tol_abs = float(np.mean(np.var(X, axis=0))) * tol
    xx = (x * x).sum(1)

    def assign(cen):
        lab = np.empty(len(x), dtype=np.int64)
        dist = np.empty(len(x))
        cc = (cen * cen).sum(1)
        for s in range(0, len(x), chunk):
            d = xx[s:s + chunk, None] - 2.0 * x[s:s + chunk] @ cen.T + cc[None, :]
            lab[s:s + chunk] = d.argmin(1)
            dist[s:s + chunk] = np.maximum(d[np.arange(d.shape[0]), lab[s:s + chunk]], 0.0)
        return lab, dist

    old = np.full(len(x), -1, dtype=np.int64)
    strict = False
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        lab, dist = assign(c)
        sums = np.zeros_like(c)
        np.add.at(sums, lab, x)
        cnt = np.bincount(lab, minlength=K).astype(np.float64)
        empty = np.nonzero(cnt == 0)[0]
        if len(empty):
            far = np.argsort(-dist, kind="stable")[:len(empty)]
            for e, f in zip(empty, far):
                sums[lab[f]] -= x[f]
                cnt[lab[f]] -= 1
                sums[e] = x[f]
                cnt[e] = 1
        new = sums / cnt[:, None]
        shift = float(((new - c) ** 2).sum())
        c = new
        if np.array_equal(lab, old):
            strict = True
            break
        if shift <= tol_abs:
            break
        old = lab
    if not strict:
        lab, dist = assign(c)
    else:
        dist = ((x - c[lab]) ** 2).sum(1)
    return c.astype(np.float32), lab, float(dist.sum()), n_iter


def dgcnnseg_features(p: Params, x: Tensor, k: int = 20, knn_fn=knn) -> Tensor:
    """get_basis.DGCNNSeg.forward(return_feat=True)[1] (get_basis.py:95-104):
    the three EdgeConv outputs concatenated, (B,192,N)."""
    ecs, _ = dgcnn_forward(p, x, k=k, training=False, prefix="encoder.", knn_fn=knn_fn)
    return torch.cat(ecs, dim=1)


def basis_flow_inputs(n_blocks: int, n_points: int, block_seed: int, label_seed: int) -> Tuple[Tensor, Tensor]:
    """Input recipe of tests/golden/basis_flow.npz (inputs are rebuilt, not stored): synthetic blocks in groups of 8 from seeds
    block_seed, block_seed + 1, ...; labels: class 1 with probability 0.93, classes 2..7 uniformly on the next 6 %,
    background (0) on the last 1 %."""
    blocks = torch.cat([synthetic_blocks(8, n_points, seed=block_seed + i) for i in range(n_blocks // 8)])
    gl = torch.Generator().manual_seed(label_seed)
    u = torch.rand(n_blocks, n_points, generator=gl)
    labels = torch.ones(n_blocks, n_points, dtype=torch.int64)
    labels[u > 0.93] = 2 + (torch.rand(n_blocks, n_points, generator=gl)[u > 0.93] * 6).long().clamp_(max=5)
    labels[u > 0.99] = 0
    return blocks, labels


def geometric_words_flow(p: Params, blocks: Tensor, labels: Tensor, num_classes: int, num_cnt: int, max_num: int,
                         rs: np.random.RandomState, k: int = 20) -> Dict[str, np.ndarray]:
    """`Get_GlobalProto_GlobalKmeans` (get_basis.py:162-222) on ONE numpy stream `rs` (the reference: numpy's global stream
    after `np.random.seed(args.seed)`, get_basis.py:310): EdgeConv features of every block in loader order (B = 1, eval mode,
    :164-170), per foreground class the labelled columns in block order (:172-183), `rs.choice(arange(n), MAX_NUM,
    replace=False)` for a class with more than MAX_NUM rows (:196-199), classes concatenated (:203-205), then
    `KMeans(num_cnt, init='k-means++').fit` (:210; sklearn centres the data and draws its seeding variates from the SAME
    stream), `Kmean2Proto` (:212) and `compute_svd` (:215).  Returns the intermediates by name."""
    feats = []
    with torch.no_grad():
        for i in range(blocks.shape[0]):
            feats.append(dgcnnseg_features(p, blocks[i:i + 1], k=k)[0])           # (192, N)
    per_class, sub_idx = [], {}
    for c in range(1, num_classes):
        cols = [f[:, labels[i] == c] for i, f in enumerate(feats) if bool((labels[i] == c).any())]
        if not cols:
            continue
        cf = torch.cat(cols, dim=1).transpose(0, 1).numpy()
        if cf.shape[0] > max_num:
            sub_idx[c] = rs.choice(np.arange(cf.shape[0]), max_num, replace=False)
            cf = cf[sub_idx[c], :]
        per_class.append(cf)
    X = np.concatenate(per_class, axis=0)
    mean = X.mean(axis=0)
    Xc = X - mean
    picks, margins = greedy_kmeans_plusplus(Xc, num_cnt, rs, with_margins=True)
    cent, lab, inertia, n_iter = sklearn_lloyd(Xc, Xc[picks])
    proto = kmean2proto(X, lab, num_cnt)
    return {"X": X, "sub_idx": sub_idx, "picks": picks, "margins": margins, "centers": cent + mean, "labels": lab,
            "inertia": inertia, "n_iter": n_iter, "proto": proto, "words": svd_reconstruct(proto)}


# --------------------------------------------------------------------------
# runs/eval.py:9-108
# --------------------------------------------------------------------------
def evaluate_metric_gfs(pred_list, gt_list, test_classes, novel_classes, learning_order, scannet=False):
    """Vectorised restatement of the per-point triple loop (eval.py:31-48) and
    the IoU bookkeeping (eval.py:50-108).  Returns (mIoU, base, novel, hm, iou[])."""
    n_cls = len(test_classes)
    lo = np.asarray(learning_order, dtype=np.int64)
    gt_c = np.zeros(n_cls, dtype=np.int64)
    pos_c = np.zeros(n_cls, dtype=np.int64)
    tp_c = np.zeros(n_cls, dtype=np.int64)
    for pr, gt in zip(pred_list, gt_list):
        pr = np.asarray(pr).astype(np.int64).ravel()
        gt = np.asarray(gt).astype(np.int64).ravel()
        gt_c += np.bincount(lo[gt], minlength=n_cls)
        pos_c += np.bincount(lo[pr], minlength=n_cls)
        tp_c += np.bincount(lo[gt][gt == pr], minlength=n_cls)
    iou = tp_c / (gt_c + pos_c - tp_c).astype(np.float64)
    first = 1 if scannet else 0
    base = [iou[c] for c in range(first, n_cls) if c not in novel_classes]
    novel = [iou[c] for c in range(first, n_cls) if c in novel_classes]
    b, n = float(np.mean(base)), float(np.mean(novel))
    return float(iou[first:].mean()), b, n, 2 * b * n / (b + n), iou[first:]


# --------------------------------------------------------------------------
# train.py harness pieces (train.py:221-305)
# --------------------------------------------------------------------------
def novel_class_coding(word_hist_per_class: Dict[int, Tensor], energy: float) -> Tensor:
    """train.py:221-238: histogram -> probability vector -> hard coding, classes sorted."""
    rows = []
    for cls in sorted(word_hist_per_class.keys()):
        h = word_hist_per_class[cls]
        rows.append(hard_coding(h / torch.sum(h), energy))
    return torch.stack(rows, dim=0)


def synthetic_blocks(B: int, N: int = 2048, seed: int = 0, dup_tail: int = 0) -> Tensor:
    """SURVEY.md §8(d) synthetic input recipe (mirrors dataloaders/loader.py:82-101):
    xyz in a 1x1x3 m block, rgb in [0,1], XYZ = xyz / per-block max -> (B,9,N)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * torch.tensor([1.0, 1.0, 3.0])
    rgb = torch.rand(B, N, 3, generator=g)
    XYZ = xyz / xyz.amax(dim=1, keepdim=True)
    x = torch.cat((xyz, rgb, XYZ), dim=2)
    if dup_tail > 0:  # loader samples with replacement when a block is short (loader.py:66)
        x[:, N - dup_tail:, :] = x[:, :dup_tail, :]
    return x.transpose(1, 2).contiguous()


# --------------------------------------------------------------------------
# dataloaders/loader.py block sampling - numpy restatement
# --------------------------------------------------------------------------
def loader_sample_indices(labels, num_point, sampled_class=0, random_sample=False, rng=None):
    """Point indices of one block (loader.py:67-82).  `rng` = numpy Generator-like with .choice (np.random by default).
    random_sample: num_point draws, with replacement iff the block is smaller.  Otherwise the sampled class keeps
    its share of the block (all of its points if the block is smaller than num_point), drawn without replacement,
    and the rest is drawn from the whole block."""
    import numpy as _np
    rng = _np.random if rng is None else rng
    n = len(labels)
    if random_sample:
        return rng.choice(_np.arange(n), num_point, replace=(n < num_point))
    valid = _np.nonzero(labels == sampled_class)[0]
    n_valid = len(valid) if n < num_point else int(len(valid) / float(n) * num_point)
    a = rng.choice(valid, n_valid, replace=False)
    b = rng.choice(_np.arange(n), num_point - n_valid, replace=(n < num_point))
    return _np.concatenate([a, b])


def loader_augment_matrix(scale=1.0, angle=0.0, mirror_x=False, mirror_y=False):
    """The 3x3 matrix augment_pointcloud builds (loader.py:132-146): uniform zoom, rotation about z, x / y mirrors
    (transforms3d.zooms.zfdir2mat / axangles.axangle2mat written out)."""
    import numpy as _np
    m = _np.eye(3) * scale
    c, s_ = _np.cos(angle), _np.sin(angle)
    m = _np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]]) @ m
    if mirror_x:
        m = _np.diag([-1.0, 1.0, 1.0]) @ m
    if mirror_y:
        m = _np.diag([1.0, -1.0, 1.0]) @ m
    return m


def loader_assemble(data, inds, pc_attribs="xyzrgbXYZ", sampled_classes=(), sampled_class=0, support=False,
                    use_all_classes=False, augm_matrix=None, jitter=None):
    """Gathered rows -> (ptcloud (n, 3 per attribute), groundtruth (n,), segment_label (n,))  (loader.py:84-129).
    augm_matrix / jitter: what augment_pointcloud draws (xyz <- xyz M^T, then + jitter), None = no augmentation."""
    import numpy as _np
    d = data[inds]
    xyz = d[:, 0:3].copy()
    rgb = d[:, 3:6]
    xyz -= _np.amin(xyz, axis=0)
    if augm_matrix is not None:
        xyz = _np.dot(xyz, _np.asarray(augm_matrix).T)
    if jitter is not None:
        xyz = xyz + _np.asarray(jitter).astype(_np.float32)
    parts = []
    if "xyz" in pc_attribs:
        parts.append(xyz)
    if "rgb" in pc_attribs:
        parts.append(rgb / 255.)
    if "XYZ" in pc_attribs:
        XYZ = xyz - _np.amin(xyz, axis=0)
        parts.append(XYZ / _np.amax(XYZ, axis=0))
    pc = _np.concatenate(parts, axis=1)
    labels = d[:, 6].astype(int)
    if support:
        gt = labels == sampled_class
    else:
        gt = _np.zeros_like(labels)
        off = 0 if use_all_classes else 1
        for i, c in enumerate(sampled_classes):
            gt[labels == c] = i + off
    seg = d[:, 7] if d.shape[1] == 8 else _np.zeros(d.shape[0], dtype=d.dtype)
    return pc, gt, seg

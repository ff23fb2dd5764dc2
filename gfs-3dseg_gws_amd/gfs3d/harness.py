"""Prototype / geometric-word coding harness, batched on the device.

Same function names, arguments and results as the helpers in the reference's
train.py:

    post_processing_hard_coding           train.py:132-152
    collect_base_class_gp_coding_sum      train.py:156-218
    collect_new_clsss_gp_coding_sum       train.py:221-238
    get_new_proto_Geo2SemProto            train.py:240-305

The reference runs three B=1 loops over whole datasets around the hot path, slicing
one-hot (K_w, N) tensors per class.  Model mode is eval (running BN statistics), so batching
changes no result; here every loader batch goes through the HIP backbone once and the per-class
word histograms are `bincount`s of (label, word) pairs on the device.  Under block-sharded data
parallelism each rank feeds its shard of the loader and the histograms are all-reduced.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from model.capl import hard_coding_mask
from . import dist as gdist


def post_processing_hard_coding(coding: torch.Tensor, energy: float) -> torch.Tensor:
    """(K_w,) frequency vector -> multi-hot of the smallest descending prefix whose mass exceeds
    energy * total; modifies and returns `coding` like the reference."""
    coding.copy_(hard_coding_mask(coding, energy))
    return coding


def _words(model, x: torch.Tensor) -> torch.Tensor:
    """(B,N) int64 geometric-word id of every point."""
    if hasattr(model, "_features"):                     # the word index straight from the assignment kernel (no one-hot tensor)
        _, _, word = model._features(x, want_semantic=False, want_onehot=False)
        return word.long()
    _, _, one_hot = model.getFeatures(x)
    return one_hot.argmax(dim=1)


@torch.no_grad()
def collect_base_class_gp_coding_sum(model, train_loader: Iterable, train_class: Sequence[int], energy: float,
                                     max_len: int = 2000):
    """-> (base_class_gp_coding (num_base, K_w) multi-hot, bg_class_coding (K_w,)).
    Loader items are (input (B,d,N), target (B,N), _) with label 0 = background and base class c
    stored as c+1 (any batch size; the reference uses 1)."""
    model.eval()
    dev = next(model.parameters()).device
    kw = model.gp.shape[0]
    n_cls = len(train_class)
    hist = torch.zeros((n_cls + 1, kw), dtype=torch.float64, device=dev)     # row = label
    bg_rows: List[torch.Tensor] = []
    for inp, target, _ in train_loader:
        inp, target = inp.to(dev), target.to(dev).long()
        if target.dim() == 1:
            target = target.unsqueeze(0)
        word = _words(model, inp)                                            # (B,N)
        flat = (target.clamp(0, n_cls) * kw + word).reshape(-1)
        keep = (target <= n_cls).reshape(-1)
        hist += torch.bincount(flat[keep], minlength=(n_cls + 1) * kw).view(n_cls + 1, kw).double()
        for b in range(inp.shape[0]):                                         # per-cloud background means
            m = target[b] == 0
            if bool(m.any()):
                bg_rows.append(torch.bincount(word[b][m], minlength=kw).float() / m.sum())
    if gdist.is_dist():
        gdist.all_reduce_counts(hist)
    rows = []
    for cls in train_class:
        h = hist[int(cls) + 1]
        rows.append(post_processing_hard_coding((h / h.sum()).float(), energy))
    base = torch.stack(rows, dim=0)
    if len(bg_rows) > max_len:
        bg_rows = random.sample(bg_rows, max_len)
    bg = torch.mean(torch.stack(bg_rows, dim=0), dim=0) if bg_rows else torch.zeros(kw, device=dev)
    return base, bg


def collect_new_clsss_gp_coding_sum(new_cls_gp_feat_dict: Dict[int, List[torch.Tensor]], energy: float) -> torch.Tensor:
    """{class: [ (1,K_w) word histograms ]} -> (n_new, K_w) multi-hot, classes in sorted order."""
    rows = []
    for cls in sorted(new_cls_gp_feat_dict.keys()):
        h = torch.sum(torch.cat(new_cls_gp_feat_dict[cls], dim=0), dim=0)
        rows.append(post_processing_hard_coding(h / torch.sum(h), energy))
    return torch.stack(rows, dim=0)


@torch.no_grad()
def get_new_proto_Geo2SemProto(val_supp_loader: Iterable, model, base_num: int = 16, novel_num: int = 5,
                               novel_class_list: Optional[Sequence[int]] = None, train_loader_NoAug=None,
                               base_class_coding=None, energy: Optional[float] = None, batch: int = 32):
    """Support clouds -> (gened_proto (n_cls,128) L2-normalised rows, novel_class_coding (n_new,K_w)).
    Loader items are (input (1,d,N), binary mask (1,N), class id (1,)).  All support clouds are
    pushed through the backbone in batches instead of one by one."""
    model.eval()
    dev = next(model.parameters()).device
    xs, ys, cs = [], [], []
    for inp, target, cls_id in val_supp_loader:
        xs.append(inp.to(dev))
        ys.append(target.to(dev))
        cs.append(int(cls_id[0]))
    feat_by_cls: Dict[int, List[torch.Tensor]] = {c: [] for c in novel_class_list}
    word_by_cls: Dict[int, List[torch.Tensor]] = {c: [] for c in novel_class_list}
    kw = model.gp.shape[0]
    for s in range(0, len(xs), batch):
        x = torch.cat(xs[s:s + batch], dim=0)
        y = torch.cat(ys[s:s + batch], dim=0)
        point_feat, _, one_hot = model.getFeatures(x)                          # (b,128,N), (b,K_w,N)
        fg = (y == 1).to(point_feat.dtype)                                     # (b,N)
        cnt = fg.sum(dim=1).clamp_min(1)
        mean_feat = torch.bmm(point_feat, fg.unsqueeze(-1)).squeeze(-1) / cnt.unsqueeze(-1)      # (b,128)
        word_hist = torch.bmm(one_hot, fg.unsqueeze(-1)).squeeze(-1)                               # (b,K_w)
        for i in range(x.shape[0]):
            c = cs[s + i]
            feat_by_cls[c].append(mean_feat[i:i + 1])
            word_by_cls[c].append(word_hist[i:i + 1])
    gened = torch.zeros_like(model.main_proto, requires_grad=False)
    gened[:base_num] = model.main_proto[:base_num].detach().clone()
    for c in novel_class_list:
        gened[c] = torch.mean(torch.cat(feat_by_cls[c], dim=0), dim=0)
    gened = F.normalize(gened, dim=1, p=2)
    coding = collect_new_clsss_gp_coding_sum(word_by_cls, energy=energy)
    return gened, coding

"""Geometric-word / prototype model with the reference's surface
(model/capl.py of Pixie8888/GFS-3DSeg_GWs): class
``mpti_net_Point_GeoAsWeight_v2`` with the same constructor, attributes
(``encoder base_learner att_learner main_proto bg_proto fusion gp``), method
names, argument meaning, return structures and state_dict keys, so train.py and
the prototype/coding harness can drive it unchanged.

Underneath, the hot path is HIP (gfs3d.ops):
  * backbone: K1 kNN + K2 EdgeConv (model/dgcnn.py here);
  * geometric-word assignment (normalise, 192x200 cosine, softmax(10.), arg-max,
    one-hot: reference capl.py:344-353) is ONE MFMA kernel;
  * get_pred (reference capl.py:290-322) is the same MFMA kernel family, logits epilogue.
Small per-class bookkeeping (<= 22 classes) stays as device tensor ops.
"""
import random

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from gfs3d import ops
from model.attention import SelfAttention
from model.dgcnn import DGCNN, BaseLearner  # noqa: F401  (BaseLearner re-exported like the reference module)

# The reference seeds the global RNGs when this module is imported (capl.py:13-17);
# a drop-in keeps that side effect so un-seeded callers get the same initial weights.
manual_seed = 321
torch.manual_seed(manual_seed)
if torch.cuda.is_available():
    torch.cuda.manual_seed_all(manual_seed)
random.seed(manual_seed)

_COS_SCALE = 10.0
_EPS = 1e-12


def _unit(t, dim):
    return t / (torch.norm(t, 2, dim, True) + _EPS)


class mpti_net_Point_GeoAsWeight_v2(nn.Module):
    """classes: total number of classes; gp: (K_w, 192) geometric words (caller-owned,
    no grad, plain attribute - not a buffer - exactly as in the reference, capl.py:49-50)."""

    def __init__(self, classes=13, criterion=nn.CrossEntropyLoss(), args=None, base_num=7, gp=None, energy=None):
        super().__init__()
        assert classes > 1
        self.criterion = criterion
        self.classes = classes
        # construction order = reference order (capl.py:38,43,44,56,57,63): same seed -> same init
        self.encoder = DGCNN(args.edgeconv_widths, args.dgcnn_mlp_widths, args.pc_in_dim, k=args.dgcnn_k,
                             return_edgeconvs=True)
        self.base_learner = BaseLearner(args.dgcnn_mlp_widths[-1], args.base_widths)
        self.att_learner = SelfAttention(args.dgcnn_mlp_widths[-1], args.output_dim)
        self.feat_dim = args.edgeconv_widths[0][-1] + args.output_dim + args.base_widths[-1]
        self.gp = gp
        self.gp.requires_grad = False
        main_dim = 128
        self.main_proto = nn.Parameter(torch.randn((classes, main_dim)))
        self.bg_proto = nn.Parameter(torch.randn((1, main_dim)))
        self.args = args
        self.base_num = base_num
        self.fusion = nn.Sequential(nn.Conv1d(self.feat_dim + self.gp.shape[0], main_dim, kernel_size=1),
                                    nn.BatchNorm1d(main_dim), nn.LeakyReLU(0.2))
        self.energy = energy
        print('model using energy: {}'.format(self.energy))

    # ------------------------------------------------------------------ features
    def _words_on(self, device):
        if self.gp.device != device:          # reference keeps gp wherever the caller put it
            self.gp = self.gp.to(device)
        return self.gp

    def getFeatures(self, x, segment_label=None):
        """x (B,C_in,N) -> point_feat (B,128,N), semantic_feat (B,192,N), one_hot (B,K_w,N)."""
        return self._features(x, want_semantic=True)

    def _heads_fusable(self):
        """The heads run on the HIP point-wise kernels (K4) when they have the reference's shapes: BaseLearner widths and the
        fusion width multiples of 64, attention head size 64, Conv1d + BatchNorm1d + LeakyReLU fusion."""
        f = self.fusion
        return (self.base_learner.fused_parts() is not None and self.att_learner.out_channel == 64
                and self.att_learner.in_channel % 64 == 0 and len(f) == 3 and isinstance(f[0], nn.Conv1d)
                and type(f[1]) in (nn.BatchNorm1d, nn.SyncBatchNorm) and isinstance(f[2], nn.LeakyReLU)
                and f[0].out_channels % 64 == 0 and f[0].in_channels <= 1024 and self.encoder.n_edgeconv <= ops.PW_MAX_SEGS
                and self.encoder.fully_fused())

    def _features(self, x, want_semantic, want_onehot=True):
        """capl.py:324-362.  Returns (point_feat, semantic or None, one-hot words - or, want_onehot=False, the (B,N) int32 word
        index).  On the GPU everything between the EdgeConv blocks and `point_feat` stays point-major and on the
        library's kernels: level-2 stack -> {q|k|v GEMM -> K5 attention, BaseLearner stack} -> one packed (B,N,512) matrix
        [soft words | edge feature | attention | base feature | zero pad] -> fusion stack (K padded from 392 to 512)."""
        words = self._words_on(x.device)
        if not (x.is_cuda and self._heads_fusable()):
            edge_feats, level2 = self.encoder(x)
            semantic = torch.cat((edge_feats[0], self.att_learner(level2), self.base_learner(level2)), dim=1)
            soft, one_hot, word = ops.WordAssignFunction.apply(torch.cat(edge_feats, dim=1), words, _COS_SCALE)
            point_feat = self.fusion(torch.cat([soft, semantic], dim=1))
            return point_feat, semantic, (one_hot if want_onehot else word)
        sync = self.encoder.bn_sync
        edge_feats, edge_pm, level2_pm = self.encoder(x, point_major=True)
        att = self.att_learner.forward_pm(level2_pm)                                        # (B,64,N)
        base = self.base_learner.forward_pm(level2_pm, sync=sync, want_cf=want_semantic)
        base_cf, base_pm = base if want_semantic else (None, base)
        conv, bn, act = self.fusion[0], self.fusion[1], self.fusion[2]
        k_in = conv.in_channels
        # K padded to a multiple of 128 (392 -> 512 for K_w = 200, 372 -> 384 for 180): the backward's input-gradient GEMM and weight
        # gradient then run their 128-wide tiles (4 column blocks instead of 7 of 64: the gradient rows are fetched once per column block)
        k_pad = (k_in + 127) // 128 * 128
        w = F.pad(conv.weight.squeeze(-1), (0, k_pad - k_in))
        edge_cat = torch.cat(edge_feats, dim=1)
        if not want_onehot and sync is None and edge_cat.shape[1] <= 192:
            # words + pack + fusion stack as ONE operator (the training / testing path: only the word index leaves it)
            point_feat, one_hot = ops.words_fusion(edge_cat, words, _COS_SCALE, [edge_pm[0], att, base_pm], [False, True, False], k_pad,
                                                   w, conv.bias, bn, self.training, float(act.negative_slope))
        else:
            if want_onehot:
                soft, one_hot, _ = ops.WordAssignFunction.apply(edge_cat, words, _COS_SCALE)
            else:
                soft, one_hot = ops.WordSoftAssignFunction.apply(edge_cat, words, _COS_SCALE)       # the word index
            # pack + fusion stack as one operator: the packed (B,N,k_pad) matrix and its gradient stay internal (16-bit in the reduced modes)
            point_feat = ops.pointwise_stack_packed([soft, edge_pm[0], att, base_pm], [True, False, True, False], k_pad,
                                                    [((w, conv.bias), bn)], self.training, float(act.negative_slope), sync=sync)
        semantic = torch.cat((edge_feats[0], att, base_cf), dim=1) if want_semantic else None
        return point_feat, semantic, one_hot

    def Get_Fg_Feat(self, x, y):
        """x (1,d,N), y (1,N) binary mask -> foreground (n,128) features and (n,K_w) one-hot words."""
        keep = y[0] == 1
        point_feat, _, word = self._features(x, want_semantic=False)
        return point_feat[0][:, keep].transpose(1, 0), word[0][:, keep].transpose(1, 0)

    # ------------------------------------------------------------------ heads
    def get_pred(self, x, proto, use_bg_proto=False):
        """10 * cosine(x[:, :, n], proto[c]) -> (B, cls(+bg), N); proto (cls,c) or (B,cls,c)."""
        if proto.dim() == 3:
            if use_bg_proto:
                proto = torch.cat([self.bg_proto.unsqueeze(0).expand(proto.shape[0], -1, -1), proto], dim=1)
        elif proto.dim() == 2:
            if use_bg_proto:
                proto = torch.cat([self.bg_proto, proto], dim=0)
        else:
            raise ValueError("proto must be (cls,c) or (b,cls,c)")
        return ops.ProtoCosineFunction.apply(x.contiguous(), proto, _COS_SCALE)

    def post_refine_proto_v2(self, proto, x, point_feat, use_bg_proto=False):
        """Eqn. 6: soft-assign points to prototypes (softmax over the POINT axis), pool
        features, blend with the prototype by their (clamped) cosine.  -> (B, classes, c)."""
        b, c, n = point_feat.shape
        resp = F.softmax(self.get_pred(x, proto, use_bg_proto), dim=2)
        pooled = torch.bmm(resp, point_feat.reshape(b, c, n).transpose(1, 2))
        if use_bg_proto:
            pooled = pooled[:, 1:, :]
        w = (F.normalize(pooled, 2, -1) * F.normalize(proto, 2, -1).unsqueeze(0)).sum(-1, keepdim=True)
        w = w * (w > 0).float()
        return w * pooled + (1 - w) * proto.unsqueeze(0)

    def get_gp_weight(self, gp_classifier, gp_feat, use_bg_weight=False, gt_label=None, th=None):
        """gp_classifier (n_cls,K_w) multi-hot coding, gp_feat (B,K_w,N) one-hot words (or their (B,N) index).
        score[b,c,n] = coding[c, word(b,n)] - a table lookup (the reference spends a
        (n_cls x K_w x N) matmul on it, capl.py:102).  Returns (weight, acc, novel_acc)."""
        word = gp_feat.long() if gp_feat.dim() == 2 else gp_feat.argmax(dim=1)     # (B,N): the word index itself, or one-hot words
        score = gp_classifier.t()[word].permute(0, 2, 1).contiguous()    # (B,n_cls,N)
        n_cls = score.shape[1]
        acc, novel_acc = 0., 0.
        if gt_label is not None and not use_bg_weight:
            hit = torch.gather(score, 1, gt_label.unsqueeze(1)).squeeze(1)            # (B,N)
            acc = hit.mean()
            novel = gt_label > self.base_num - 1
            novel_acc = hit[novel].mean() if bool(novel.any()) else torch.zeros_like(acc)
        elif gt_label is not None and use_bg_weight:
            fg = gt_label > 0
            hit = torch.gather(score, 1, (gt_label - 1).clamp_min(0).unsqueeze(1)).squeeze(1) * fg
            acc = hit.mean()
        weight = torch.where(score == 1, torch.as_tensor(th, dtype=score.dtype, device=score.device),
                             torch.ones((), dtype=score.dtype, device=score.device))
        if use_bg_weight:
            weight = torch.cat([torch.ones_like(weight[:, :1]), weight], dim=1)
            weight.scatter_(1, gt_label.unsqueeze(1), 1.0)
        return weight, acc, novel_acc

    # ------------------------------------------------------------------ forward
    def forward(self, x, y=None, gened_proto=None, gen_proto=False, eval_model=False, target_cls=None,
                segment_label=None, geo2sem_proto=None, base_class_coding=None, novel_class_coding=None,
                bg_class_coding=None):
        """Training: returns (pred (B,N), loss).  eval_model=True: returns
        (logits (B,cls,N) re-weighted by the geometric-word coding, gp_acc, gp_novel_acc)."""
        nb = self.base_num
        half = x.size(0) // 2
        fake_novel = None
        pending = None
        if not eval_model:
            # The fake-novel draw needs the set of classes present in the support half ON THE HOST (python RNG, as the
            # reference).  The presence kernel and its 8-byte copy are queued BEFORE the backbone and read AFTER the backbone has
            # been queued: the host waits for one tiny kernel on the labels while the device has the whole forward pass to run
            # (waiting right here left the device idle for ~0.15 ms per step while the host reacted; the same draw from the
            # same RNG state: nothing consumes python's `random` in between).
            pending = self._present_begin(y[half:])
            if pending is None:
                fake_novel = self._draw_fake_novel(y[half:])
        point_feat, _, word = self._features(x, want_semantic=False, want_onehot=False)      # word: (B,N) index
        if pending is not None:
            fake_novel = self._draw_fake_novel(y[half:], present=self._present_end(pending))

        if eval_model:
            if gened_proto.dim() == 3:
                gened_proto = gened_proto[0]
            coding = torch.cat([base_class_coding, novel_class_coding], dim=0)
            if (self.fuse_head and not torch.is_grad_enabled() and word.dtype == torch.int32 and coding.shape[0] == self.main_proto.shape[0]
                    and ops.capl_head_supported(point_feat, coding.shape[0], 0) and coding.numel() <= 16384
                    and (y is None or (y.is_cuda and y.dtype == torch.int64))):
                # the whole testing head (refinement, get_pred, geometric-word re-weighting, accuracy figures): csrc/head.hip
                return ops.capl_eval_head(point_feat, self.main_proto.detach(), gened_proto.to(point_feat.dtype), word, coding.to(point_feat.dtype),
                                          self.args.eval_weight, nb, y)
            refined = self.post_refine_proto_v2(proto=self.main_proto, x=point_feat, point_feat=point_feat)
            refined = torch.cat([refined[:, :nb] + gened_proto[:nb].unsqueeze(0),
                                 refined[:, nb:] * 0 + gened_proto[nb:].unsqueeze(0)], dim=1)
            logits = self.get_pred(point_feat, refined)
            weight, gp_acc, gp_novel_acc = self.get_gp_weight(coding, word, gt_label=y, th=self.args.eval_weight)
            return logits * weight, gp_acc, gp_novel_acc

        return self.train_head(point_feat, y, fake_novel)

    def train_head(self, point_feat, y, fake_novel):
        """capl.py:194-242 after getFeatures: (pred (B,N), loss).  One fused HIP pipeline, forward and backward
        (csrc/head.hip: fake prototypes, refinement, both cosine heads, both cross-entropies, arg-max) when the criterion is the
        reference's CrossEntropyLoss; the tensor-library form below otherwise (and as the fused form's reference in the tests)."""
        nb = self.base_num
        half = point_feat.size(0) // 2
        if self._head_fusable(point_feat, y, fake_novel):
            return ops.capl_train_head(point_feat, self.main_proto, self.bg_proto, y, fake_novel, nb, self.criterion.ignore_index)
        ori_proto, _ = self.generate_fake_proto(x=point_feat[half:], y=y[half:], main_proto=self.main_proto.clone(),
                                                fake_novel=fake_novel)
        loss_1 = self.criterion(self.get_pred(point_feat, ori_proto, use_bg_proto=True), y)
        refined = self.post_refine_proto_v2(proto=self.main_proto.clone(), x=point_feat, point_feat=point_feat,
                                            use_bg_proto=True)
        refined = torch.cat([refined[:, :nb] + ori_proto[:nb].unsqueeze(0),
                             refined[:, nb:] * 0 + ori_proto[nb:].unsqueeze(0)], dim=1)
        logits_2 = self.get_pred(point_feat, refined, use_bg_proto=True)
        loss_2 = self.criterion(logits_2, y)
        return logits_2.max(1)[1], 0.5 * loss_2 + 0.5 * loss_1

    fuse_head = True          # False: the unfused (tensor-library) form of the training head, kept as the reference of the fused one

    def _head_fusable(self, point_feat, y, fake_novel):
        """The fused training head reproduces `criterion` only when it is the reference's plain CrossEntropyLoss
        (train.py:317: mean reduction, no class weights, no label smoothing, any ignore_index)."""
        crit = self.criterion
        return (self.fuse_head and type(crit) is nn.CrossEntropyLoss and crit.weight is None and crit.reduction == 'mean'
                and getattr(crit, 'label_smoothing', 0.0) == 0.0 and y is not None and y.dtype == torch.int64 and y.is_cuda
                and ops.capl_head_supported(point_feat, self.main_proto.shape[0], len(fake_novel))
                and self.main_proto.shape[1] == point_feat.shape[1])

    # ------------------------------------------------------------------ prototypes
    def generate_fake_proto(self, x, y, main_proto, fake_novel=None, post_processing=False):
        """Training only: pick half of the foreground classes present in the support half
        of the batch as 'fake novel' (python RNG, as the reference) and replace their
        prototype by the L2-normalised masked feature mean.  Returns (proto, fake_novel)."""
        labels = y.unsqueeze(1)
        dp = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if fake_novel is None:
            fake_novel = self._draw_fake_novel(y, n_cls=main_proto.size(0))
        proto = _unit(main_proto, 1)
        feat = _unit(x, 1)
        fake = [int(c) for c in fake_novel]
        if fake:
            # all fake-novel classes at once (the reference loops over them, capl.py:392-408): masked sums as one
            # batched GEMM against the (b, F, n) class masks, rows replaced by index_copy (== proto (1 - pick) + mean pick)
            cls_idx = torch.tensor(fake, dtype=torch.long, device=feat.device)
            m = (labels == cls_idx.view(1, -1, 1)).to(feat.dtype)              # (b, F, n)
            num = torch.matmul(feat, m.transpose(1, 2)).sum(0).t()             # (F, C)
            den = m.sum((0, 2))                                                # (F,)
            if dp:   # masked mean over the GLOBAL support half: differentiable SUM all-reduce of (sum, count)
                from gfs3d.dist import all_reduce_sum_diff as _ar
                packed = _ar(torch.cat([num, den.unsqueeze(1)], dim=1))
                num, den = packed[:, :-1], packed[:, -1]
            mean = num / (den + _EPS).unsqueeze(1)
            if post_processing:
                mean = torch.stack([self.post_processing_hard_coding(row) for row in mean.unbind(0)])
            proto = proto.index_copy(0, cls_idx - 1, mean)
        return proto, fake_novel

    def _present_begin(self, y):
        """Queue the label presence of y for a later _present_end(); None when the synchronous path has to be taken
        (block-sharded batch: a collective; labels not on the fused kernel's path)."""
        dp = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if dp or not (y.is_cuda and y.dtype == torch.int64 and self.main_proto.size(0) < 64):
            return None
        return ops.label_presence_begin(y)

    def _present_end(self, pending):
        n_cls = self.main_proto.size(0)
        return [v for v in ops.label_presence_end(pending) if v != 0 and v <= n_cls]

    def _draw_fake_novel(self, y, n_cls=None, present=None):
        """Half of the foreground classes present in y (python RNG, reference capl.py:384-390).  Block-sharded batch: the
        class set is the union over ranks (one MAX all-reduce of a presence vector); every rank then draws the SAME sample
        (python RNG seeded alike on all ranks, as train.py does)."""
        n_cls = self.main_proto.size(0) if n_cls is None else n_cls
        dp = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if present is not None:
            pass
        elif dp:
            seen = torch.zeros(n_cls + 1, dtype=torch.int32, device=y.device)
            seen[y.unique().long().clamp_max(n_cls)] = 1
            torch.distributed.all_reduce(seen, op=torch.distributed.ReduceOp.MAX)
            present = [int(v) for v in torch.nonzero(seen).flatten().tolist() if int(v) != 0]
        elif y.is_cuda and y.dtype == torch.int64 and n_cls < 64:
            # ascending like unique(); labels beyond the classes (an ignore label: the reference's sample would index out of range) are skipped
            present = [v for v in ops.label_presence(y) if v != 0 and v <= n_cls]
        else:
            present = [int(v) for v in y.unique().tolist() if int(v) != 0]
        return random.sample(present, len(present) // 2)

    def post_processing_hard_coding(self, coding):
        """(K_w,) probability vector -> multi-hot of the smallest descending prefix whose
        mass exceeds energy * total (in place, like the reference)."""
        mask = hard_coding_mask(coding, self.energy)
        coding.copy_(mask)
        return coding


def hard_coding_mask(coding, energy):
    """Sequential fp32 prefix sum on the host (<= a few hundred entries): identical
    accumulation order to the reference loop (capl.py:419-428, train.py:132-152)."""
    v = coding.detach().float().cpu()
    order = torch.argsort(v, descending=True)
    csum = np.cumsum(v[order].numpy(), dtype=np.float32)
    thr = np.float32(energy) * np.float32(v.sum().item())
    over = np.nonzero(csum > thr)[0]
    keep = len(order) if over.size == 0 else int(over[0]) + 1
    mask = torch.zeros_like(v)
    mask[order[:keep]] = 1.0
    return mask.to(device=coding.device, dtype=coding.dtype)

"""DGCNN backbone with the reference's module surface (model/dgcnn.py of
Pixie8888/GFS-3DSeg_GWs): ``knn``, ``get_edge_feature``, ``conv2d``, ``conv1d``,
``DGCNN``, ``BaseLearner``, ``DGCNNSeg_att`` - same constructor arguments,
forward() contracts and state_dict keys (``edge_convs.{i}.layer.{0,1,3,4}.*``,
``conv.layer.{0,1,3,4}.*``), so reference checkpoints load and train.py /
get_basis.py can import it unchanged.

What is different is everything underneath: kNN graph construction and each
EdgeConv block run as hand-written HIP kernels (gfs3d.ops -> libgfs3d_hip.so);
no (B,N,N) distance matrix and no (B,2C,N,k) / (B,64,N,k) edge tensor exists.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from gfs3d import ops
from model.attention import SelfAttention

_LRELU = 0.2


def knn(x, k):
    """(B,C,N) -> (B,N,k) int64 neighbour indices, nearest first (self included).
    Reference: dgcnn.py:17-23.  Fused distance + top-k on the GPU."""
    return ops.knn_topk(x, k, want_i64=True, want_i32=False)


def get_edge_feature(x, K=20, idx=None):
    """(B,C,N) -> (B,2C,N,K) = cat(x_j - x_i, x_i) (reference dgcnn.py:26-42).
    Kept for API compatibility; DGCNN.forward below never materialises it."""
    B, C, N = x.shape
    if idx is None:
        idx = knn(x, K)
    nbr = torch.gather(x, 2, idx.reshape(B, 1, N * K).expand(B, C, N * K)).view(B, C, N, K)
    ctr = x.unsqueeze(-1).expand(B, C, N, K)
    return torch.cat((nbr - ctr, ctr), dim=1)


def _pointwise_stack(conv_cls, bn_cls, in_feat, layer_dims, batch_norm, relu, bias):
    mods, prev = [], in_feat
    for width in layer_dims:
        mods.append(conv_cls(prev, width, kernel_size=1, bias=bias))
        if batch_norm:
            mods.append(bn_cls(width))
        if relu:
            mods.append(nn.LeakyReLU(_LRELU))
        prev = width
    return nn.Sequential(*mods)


class conv2d(nn.Module):
    """[Conv2d 1x1 -> BatchNorm2d -> LeakyReLU(0.2)] per entry of layer_dims, held in
    ``self.layer`` (reference dgcnn.py:45-61)."""

    def __init__(self, in_feat, layer_dims, batch_norm=True, relu=True, bias=False):
        super().__init__()
        self.layer_dims = layer_dims
        self.layer = _pointwise_stack(nn.Conv2d, nn.BatchNorm2d, in_feat, layer_dims, batch_norm, relu, bias)

    def forward(self, x):
        return self.layer(x)

    def fused_parts(self):
        """(conv1, bn1, conv2, bn2) when this block has the shape the HIP EdgeConv
        kernel implements (two bias-free 1x1 convs to 64 channels with BN + LeakyReLU)."""
        m = list(self.layer)
        bn_types = (nn.BatchNorm2d, nn.SyncBatchNorm)        # convert_sync_batchnorm() keeps weights/buffers
        ok = (len(m) == 6 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], bn_types)
              and isinstance(m[3], nn.Conv2d) and isinstance(m[4], bn_types)
              and m[0].bias is None and m[3].bias is None
              and m[0].out_channels == ops.EC_WIDTH and m[3].out_channels == ops.EC_WIDTH
              and m[1].track_running_stats and m[4].track_running_stats and m[1].affine and m[4].affine)
        if not ok:
            raise NotImplementedError(
                "gfs3d: the HIP EdgeConv kernel implements edgeconv widths [64, 64] with BatchNorm + LeakyReLU "
                f"(got layer_dims={self.layer_dims}); DGCNN.forward runs other shapes through the general path")
        return m[0], m[1], m[3], m[4]

    def is_fused_shape(self):
        try:
            self.fused_parts()
            return True
        except NotImplementedError:
            return False


class conv1d(nn.Module):
    """Point-wise MLP [Conv1d 1x1 -> BatchNorm1d -> LeakyReLU(0.2)] (reference dgcnn.py:64-80)."""

    def __init__(self, in_feat, layer_dims, batch_norm=True, relu=True, bias=False):
        super().__init__()
        self.layer_dims = layer_dims
        self.layer = _pointwise_stack(nn.Conv1d, nn.BatchNorm1d, in_feat, layer_dims, batch_norm, relu, bias)

    def forward(self, x):
        return self.layer(x)

    def fused_parts(self):
        """[(conv, bn)] when the stack has the shape the HIP point-wise kernels implement (bias-free 1x1 convs
        to multiples of 64 channels, BatchNorm1d, LeakyReLU(0.2); at most 4 layers) - else None."""
        m = list(self.layer)
        if len(m) % 3 or not 3 <= len(m) <= 12:
            return None
        parts = []
        for i in range(0, len(m), 3):
            c, bn, act = m[i], m[i + 1], m[i + 2]
            # (convert_sync_batchnorm() swaps the class and keeps weights / buffers: the HIP stack then shares its statistics
            # through gfs3d.ops.BNSync instead)
            ok = (isinstance(c, nn.Conv1d) and type(bn) in (nn.BatchNorm1d, nn.SyncBatchNorm) and isinstance(act, nn.LeakyReLU)
                  and c.bias is None and c.kernel_size == (1,) and c.out_channels % 64 == 0 and c.out_channels <= 1024
                  and bn.affine and bn.track_running_stats and abs(act.negative_slope - _LRELU) < 1e-12)
            if not ok:
                return None
            parts.append((c, bn))
        return parts


class DGCNN(nn.Module):
    """Stacked EdgeConv blocks (kNN graph rebuilt on every block's features) followed by
    a point-wise MLP on the concatenated block outputs (reference dgcnn.py:83-127)."""

    def __init__(self, edgeconv_widths, mlp_widths, nfeat, k=20, return_edgeconvs=False):
        super().__init__()
        self.n_edgeconv = len(edgeconv_widths)
        self.k = k
        self.return_edgeconvs = return_edgeconvs
        self.edge_convs = nn.ModuleList()
        prev = nfeat
        for widths in edgeconv_widths:
            self.edge_convs.append(conv2d(2 * prev, widths))
            prev = widths[-1]
        self.conv = conv1d(sum(w[-1] for w in edgeconv_widths), mlp_widths)
        self.bn_sync = None      # gfs3d.ops.BNSync: whole-batch BatchNorm statistics under block-sharded DP

    def fully_fused(self):
        """Every block has the fused EdgeConv shape and the point-wise stack the K4 form: forward(point_major=True) is available
        (training mode, or inference without gradients)."""
        parts = self.conv.fused_parts() if self.n_edgeconv <= ops.PW_MAX_SEGS else None
        return (all(b.is_fused_shape() for b in self.edge_convs) and self.k <= ops.MAX_K and parts is not None
                and parts[0][0].in_channels == ops.PW_SEG_WIDTH * self.n_edgeconv
                and (self.training or not torch.is_grad_enabled()))

    def edgeconv_features(self, x, with_point_major=False, also_track=()):
        """The n_edgeconv block outputs, each (B,64,N).  HIP path: K1 (kNN) + K2 (EdgeConv).
        with_point_major: also the (B,N,64) copies the kernels keep (input of the point-wise stack, K4).
        also_track: further BatchNorm modules whose `num_batches_tracked` rides in this call's one multi-tensor add."""
        feats, feats_pm, bns = [], [], list(also_track)
        x_pm = x_sq = x_img = None
        for li, block in enumerate(self.edge_convs):
            # Shapes outside the fused kernel (reference conv2d accepts any layer_dims, dgcnn.py:45-58; k > 32; a backward pass in
            # eval mode): the graph still comes from the HIP kNN kernel (bit-exact contract), the block itself runs as the
            # reference wrote it - gather, cat, the nn.Sequential, max over k - as tensor-library ops on the device.
            if (not block.is_fused_shape() or self.k > ops.MAX_K or x.shape[1] > 128
                    or (not self.training and torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in block.parameters())))):
                if self.bn_sync is not None and self.training:
                    raise NotImplementedError("gfs3d: synchronised BatchNorm is implemented for the fused EdgeConv shape ([64, 64], k <= 32) only")
                x = block(get_edge_feature(x, self.k, ops.knn_topk(x.detach(), self.k))).max(dim=-1)[0]
                x_pm = x_sq = x_img = None
                feats.append(x)
                feats_pm.append(None)
                continue
            c1, bn1, c2, bn2 = block.fused_parts()
            pm_ok = x_pm is not None and x.shape[1] == 64
            idx32 = ops.knn_topk(x.detach(), self.k, want_i64=False, want_i32=True, sqnorm=x_sq,
                                 x_pm=x_pm.detach() if pm_ok else None, x_img=x_img if pm_ok else None)
            # the block's epilogue leaves what the NEXT block's kNN needs: norms, point-major rows and the screen's operand rows
            nxt = li + 1 < len(self.edge_convs)
            x, x_pm, x_sq, x_img = ops.edgeconv_block(x, idx32, c1.weight, bn1, c2.weight, bn2, self.training, _LRELU,
                                                      sync=self.bn_sync, x_pm=x_pm, track=False, want_sq=True, want_img=True) \
                if nxt else ops.edgeconv_block(x, idx32, c1.weight, bn1, c2.weight, bn2, self.training, _LRELU,
                                               sync=self.bn_sync, x_pm=x_pm, track=False, want_sq=True) + (None,)
            bns += [bn1, bn2]
            feats.append(x)
            feats_pm.append(x_pm)
        if self.training:               # nn.BatchNorm bookkeeping of all blocks in one multi-tensor add
            counters = [bn.num_batches_tracked for bn in bns if bn.num_batches_tracked is not None]
            if counters:
                torch._foreach_add_(counters, 1)
        return (feats, feats_pm) if with_point_major else feats

    def forward(self, x, point_major=False):
        """point_major=True (the fused CAPL heads): returns (block outputs, their (B,N,64) copies, level-2 feature as a
        point-major (B,N,C) tensor) - no channel-first copy of the level-2 feature is written."""
        parts = self.conv.fused_parts() if self.n_edgeconv <= ops.PW_MAX_SEGS else None
        # the point-wise stack's BatchNorm counters ride in the EdgeConv blocks' multi-tensor add when the stack will run fused
        # (decided below from the same `parts`; a stack that turns out not to be fused bumps its own counters as nn.BatchNorm1d)
        will_fuse = parts is not None and parts[0][0].in_channels == ops.PW_SEG_WIDTH * self.n_edgeconv and self.training and x.is_cuda \
            and self.fully_fused()
        feats, feats_pm = self.edgeconv_features(x, with_point_major=True, also_track=[bn for _c, bn in parts] if will_fuse else ())
        fused = (parts is not None and parts[0][0].in_channels == ops.PW_SEG_WIDTH * len(feats) and all(t is not None for t in feats_pm)
                 and (self.training or not (torch.is_grad_enabled() and any(t.requires_grad for t in feats))))
        if will_fuse and not fused:      # (a block took the general path after all: nn.BatchNorm1d will count this step itself)
            counters = [bn.num_batches_tracked for _c, bn in parts if bn.num_batches_tracked is not None]
            if counters:                 # (track_running_stats=False: nothing was bumped)
                torch._foreach_sub_(counters, 1)
            will_fuse = False
        if point_major:
            if not fused:
                raise NotImplementedError("DGCNN(point_major=True): the point-wise stack is not in the HIP-fusable form")
            out_pm = ops.pointwise_stack(feats_pm, parts, self.training, _LRELU, sync=self.bn_sync, want_cf=False, want_pm=True,
                                         track=not will_fuse)
            return feats, feats_pm, out_pm
        if fused:
            out = ops.pointwise_stack(feats_pm, parts, self.training, _LRELU, sync=self.bn_sync, track=not will_fuse)       # K4: no cat, no transposes
        else:
            out = self.conv(torch.cat(feats, dim=1))
        if self.return_edgeconvs:
            return feats, out
        return feats[0], out


class BaseLearner(nn.Module):
    """Conv1d(bias) + BatchNorm1d stack with ReLU between (not after) the layers
    (reference dgcnn.py:130-152); keys ``convs.{i}.{0,1}.*``."""

    def __init__(self, in_channels, params):
        super().__init__()
        self.num_convs = len(params)
        dims = [in_channels] + list(params)
        self.convs = nn.ModuleList(
            nn.Sequential(nn.Conv1d(dims[i], dims[i + 1], 1), nn.BatchNorm1d(dims[i + 1]))
            for i in range(self.num_convs))

    def forward(self, x):
        last = self.num_convs - 1
        for i, blk in enumerate(self.convs):
            x = blk(x)
            if i != last:
                x = F.relu(x)
        return x

    def fused_parts(self):
        """[(conv, bn)] when the HIP point-wise stack implements this learner (1x1 convolutions to multiples of 64
        channels from a multiple of 64, BatchNorm1d with running statistics; at most 4 layers) - else None."""
        if not 1 <= self.num_convs <= ops.PW_MAX_LAYERS:
            return None
        parts = []
        for blk in self.convs:
            c, bn = blk[0], blk[1]
            ok = (isinstance(c, nn.Conv1d) and type(bn) in (nn.BatchNorm1d, nn.SyncBatchNorm) and c.kernel_size == (1,)
                  and c.in_channels % 64 == 0 and c.in_channels <= 1024 and c.out_channels % 64 == 0
                  and c.out_channels <= 1024 and bn.affine and bn.track_running_stats)
            if not ok:
                return None
            parts.append((c, bn))
        return parts

    def forward_pm(self, x_pm, sync=None, want_cf=False):
        """x_pm (B, N, C) point-major -> (B, N, C_out) point-major [and (B, C_out, N)]: the whole learner as ONE K4 stack
        (ReLU = LeakyReLU(0) between the layers, slope 1 = no activation after the last; the conv biases only move the
        running means)."""
        parts = self.fused_parts()
        if parts is None:
            raise NotImplementedError("BaseLearner.forward_pm: widths outside the HIP point-wise stack")
        slopes = [0.0] * (self.num_convs - 1) + [1.0]
        return ops.pointwise_stack([x_pm], parts, self.training, 0.0, sync=sync, slopes=slopes, want_cf=want_cf, want_pm=True)


class DGCNNSeg_att(nn.Module):
    """Pre-train style segmenter (reference dgcnn.py:155-202): encoder + base learner +
    attention -> point-wise classifier."""

    def __init__(self, args, num_classes):
        super().__init__()
        self.encoder = DGCNN(args.edgeconv_widths, args.dgcnn_mlp_widths, args.pc_in_dim, k=args.dgcnn_k)
        self.base_learner = BaseLearner(args.dgcnn_mlp_widths[-1], args.base_widths)
        self.att_learner = SelfAttention(args.dgcnn_mlp_widths[-1], args.output_dim)
        self.feat_dim = args.edgeconv_widths[0][-1] + args.output_dim + args.base_widths[-1]
        self.segmenter = nn.Sequential(
            nn.Conv1d(self.feat_dim, 256, 1, bias=False), nn.BatchNorm1d(256), nn.LeakyReLU(_LRELU),
            nn.Conv1d(256, 128, 1), nn.BatchNorm1d(128), nn.LeakyReLU(_LRELU),
            nn.Dropout(0.3), nn.Conv1d(128, num_classes, 1))

    def forward(self, pc, return_feat=False):
        if pc.is_cuda and segmenter_fusable(self.segmenter) and self.base_learner.fused_parts() is not None \
                and self.att_learner.out_channel == 64 and self.encoder.fully_fused():
            # everything after the EdgeConv blocks on the HIP point-wise kernels, point-major throughout
            feats, feats_pm, level2_pm = self.encoder(pc, point_major=True)
            att = self.att_learner.forward_pm(level2_pm)
            base_pm = self.base_learner.forward_pm(level2_pm, sync=self.encoder.bn_sync)
            packed = ops.pack_rows([feats_pm[0], att, base_pm], [False, True, False])
            logits = segmenter_forward_pm(self.segmenter, packed, self.training, self.encoder.bn_sync)
            return (logits, feats[0]) if return_feat else logits
        level1, level2 = self.encoder(pc)
        feat = torch.cat((level1, self.att_learner(level2), self.base_learner(level2)), dim=1)
        logits = self.segmenter(feat)
        return (logits, level1) if return_feat else logits


def segmenter_fusable(seg):
    """The point-wise classifier of the pre-training models (reference dgcnn.py:170-177, get_basis.py:84-93):
    [Conv1d, BatchNorm1d, LeakyReLU] x 2, Dropout, Conv1d - the two BatchNorm layers run as one K4 stack."""
    m = list(seg)
    if len(m) != 8:
        return False
    c1, b1, a1, c2, b2, a2, dr, c3 = m
    return (isinstance(c1, nn.Conv1d) and isinstance(c2, nn.Conv1d) and isinstance(c3, nn.Conv1d) and isinstance(dr, nn.Dropout)
            and type(b1) in (nn.BatchNorm1d, nn.SyncBatchNorm) and type(b2) in (nn.BatchNorm1d, nn.SyncBatchNorm)
            and isinstance(a1, nn.LeakyReLU) and isinstance(a2, nn.LeakyReLU)
            and c1.in_channels % 64 == 0 and c1.in_channels <= 1024 and c1.out_channels % 64 == 0 and c2.out_channels % 64 == 0
            and c1.kernel_size == (1,) and c2.kernel_size == (1,))


def segmenter_forward_pm(seg, packed_pm, training, sync=None):
    """packed_pm (B, N, C_in) point-major -> logits (B, num_classes, N)."""
    c1, b1, a1, c2, b2, a2, dr, c3 = list(seg)
    slopes = [float(a1.negative_slope), float(a2.negative_slope)]
    if c3.kernel_size == (1,) and c3.in_channels % 64 == 0 and c3.in_channels <= 1024 and c3.out_channels <= 64:
        # the classifier (reference dgcnn.py:176-177 / get_basis.py:92-93: Dropout, Conv1d(128, classes)) on the library's GEMM
        # too: the stack hands its output over point-major, the weight is padded to one 64-row tile
        h_pm = ops.pointwise_stack([packed_pm], [(c1, b1), (c2, b2)], training, _LRELU, sync=sync, slopes=slopes,
                                   want_cf=False, want_pm=True)
        w = F.pad(c3.weight[:, :, 0], (0, 0, 0, 64 - c3.out_channels))
        logits = ops.linear_cf(dr(h_pm), w)[0][:, :c3.out_channels]
        return logits + c3.bias[None, :, None] if c3.bias is not None else logits
    h = ops.pointwise_stack([packed_pm], [(c1, b1), (c2, b2)], training, _LRELU, sync=sync, slopes=slopes)
    return c3(dr(h))

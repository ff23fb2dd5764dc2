"""Host-side operators over the C ABI (include/gfs3d.h).

torch is plumbing here: device memory (torch.empty), the current HIP stream
and autograd bookkeeping.  All arithmetic of the hot path happens inside
libgfs3d_hip.so.  Every operator insists on CUDA(HIP) fp32 tensors and raises
otherwise - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import PointwiseArgs, PointwiseBwdArgs, PW_MAX_LAYERS, PW_MAX_SEGS, PW_SEG_WIDTH, EdgeConvArgs, EdgeConvBwdArgs

EC_WIDTH = 64
MAX_K = 32

# --------------------------------------------------------------------------
# arithmetic of the GEMM-shaped kernels (include/gfs3d.h: GFS3D_COMPUTE_*)
# --------------------------------------------------------------------------
_COMPUTE_IDS = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f16": _lib.COMPUTE_F16}
_compute = _lib.COMPUTE_F32


def set_compute(mode: str) -> str:
    """Select the matrix-core operand form of every following call: "f32" (default: fp32 accuracy, exact 3-way bf16
    split), "bf16" (one bf16 product per fp32 product) or "f16" (half operands forward, bf16 backward) - BASELINE
    configs[4]'s reduced-precision mode.  Tensors stay fp32; kNN, BatchNorm statistics and reductions are unaffected.
    A backward pass uses the mode its forward ran in.  Returns the previous mode."""
    global _compute
    if mode not in _COMPUTE_IDS:
        raise ValueError(f"gfs3d.set_compute: mode must be one of {sorted(_COMPUTE_IDS)}, got {mode!r}")
    prev = get_compute()
    _compute = _COMPUTE_IDS[mode]
    return prev


def get_compute() -> str:
    return next(k for k, v in _COMPUTE_IDS.items() if v == _compute)


_store16 = __import__("os").environ.get("GFS3D_STORE16", "1") != "0"      # (GFS3D_STORE16=0: A/B runs, scripts/ab_env.sh)


def set_store16(on: bool) -> bool:
    """Reduced modes only: keep the point-wise stacks' internal tensors (pre-BatchNorm activations, their gradients, the packed
    fusion input) in 16 bits in HBM (include/gfs3d.h: gfs3d_pointwise_args.store16) - default on; off = fp32 storage with
    16-bit matrix-core operands only (the form of rounds 3-5, kept for A/B measurements).  Returns the previous setting."""
    global _store16
    prev, _store16 = _store16, bool(on)
    return prev


def _h_dtype(compute: int, training: bool):
    """Element type of a stack's internal activations for a call in mode `compute`."""
    if not (_store16 and training) or compute == _lib.COMPUTE_F32:
        return torch.float32
    return torch.float16 if compute == _lib.COMPUTE_F16 else torch.bfloat16


def set_deterministic(on: bool) -> None:
    """Bit-reproducible EdgeConv forward / backward (include/gfs3d.h: gfs3d_set_deterministic): the BatchNorm reductions
    are added in a fixed order instead of with float64 atomics.  Process-wide."""
    _lib.check(_lib.lib().gfs3d_set_deterministic(1 if on else 0), "gfs3d_set_deterministic")


class compute:
    """`with ops.compute("bf16"): ...` - scoped set_compute()."""

    def __init__(self, mode: str):
        self.mode = mode

    def __enter__(self):
        self.prev = set_compute(self.mode)
        return self

    def __exit__(self, *exc):
        set_compute(self.prev)
        return False


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"gfs3d: `{name}` must be a HIP (cuda) tensor - the hot path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"gfs3d: `{name}` must be float32, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_workspaces = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """One growing scratch buffer per device; all launches are on the current
    stream, so consecutive calls reuse it safely."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# --------------------------------------------------------------------------
# K1
# --------------------------------------------------------------------------
def knn_topk(x: torch.Tensor, k: int, want_i64: bool = True, want_i32: bool = False, sqnorm: Optional[torch.Tensor] = None,
             x_pm: Optional[torch.Tensor] = None, x_img: Optional[torch.Tensor] = None):
    """x (B,C,N) channel-first -> neighbour indices (B,N,k), nearest first.
    Replaces knn() of model/dgcnn.py:17-23.  `sqnorm`: the point norms if the caller has them in the kernel's
    summation order (the third output of `edgeconv_block(..., want_sq=True)` for the previous block's features);
    `x_pm`: the (B,N,64) point-major values of x when C == 64 (the previous block's second output) - the screened
    path's exact pass reads point rows; `x_img`: the screen's operand rows of the same features (the fourth output of
    `edgeconv_block(..., want_img=True)`; used together with x_pm)."""
    _need_gpu(x, "x")
    if x.dim() != 3:
        raise RuntimeError("gfs3d.knn_topk: x must be (B,C,N)")
    x = x.contiguous()
    B, C, N = x.shape
    i64 = torch.empty((B, N, k), dtype=torch.int64, device=x.device) if want_i64 else None
    i32 = torch.empty((B, N, k), dtype=torch.int32, device=x.device) if want_i32 else None
    if sqnorm is not None and (sqnorm.dtype != torch.float32 or not sqnorm.is_contiguous() or sqnorm.numel() < B * N + 16
                               or sqnorm.device != x.device):
        raise RuntimeError("gfs3d.knn_topk: sqnorm must be a contiguous fp32 tensor of B*N + 16 elements on x's device")
    if x_pm is not None and (C != 64 or tuple(x_pm.shape) != (B, N, 64) or not x_pm.is_contiguous() or x_pm.dtype != torch.float32):
        raise RuntimeError("gfs3d.knn_topk: x_pm must be the contiguous (B,N,64) fp32 point-major copy of a 64-channel x")
    if x_img is not None and (x_pm is None or x_img.dtype != torch.int32 or not x_img.is_contiguous() or x_img.device != x.device
                              or x_img.numel() < (B * N + _lib.KNN_IMG_SLACK_ROWS) * 64):
        raise RuntimeError("gfs3d.knn_topk: x_img must be the int32 operand image (B*N + 128 rows of 64 dwords) that came with x_pm")
    with torch.cuda.device(x.device):
        L = _lib.lib()
        ws = _workspace(x.device, L.gfs3d_knn_workspace_bytes(B, C, N, int(k)))
        rc = L.gfs3d_knn_topk_img_f32(_p(x), _p(x_pm), _p(x_img), _p(sqnorm), B, C, N, int(k), _p(i64), _p(i32), _p(ws), ws.numel(),
                                      _stream())
    _lib.check(rc, "gfs3d_knn_topk_img_f32")
    if want_i64 and want_i32:
        return i64, i32
    return i64 if want_i64 else i32


# --------------------------------------------------------------------------
# K2
# --------------------------------------------------------------------------
class BNSync:
    """How BatchNorm statistics are shared between the ranks of a block-sharded batch: `all_reduce`
    sums a (128,) float64 device tensor in place over the ranks, `world_size` scales the sample
    count (equal shards).  `BNSync.from_group()` wraps torch.distributed (RCCL)."""

    def __init__(self, all_reduce, world_size: int):
        self.all_reduce, self.world_size = all_reduce, int(world_size)

    @staticmethod
    def from_group(group=None):
        import torch.distributed as dist
        return BNSync(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group), dist.get_world_size(group))


class EdgeConvRun:
    """One EdgeConv block invocation: owns the buffers the forward leaves for the backward and drives
    the C ABI either monolithically or phase by phase (synchronised BatchNorm)."""

    def __init__(self, x, idx, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, training, eps, momentum, slope,
                 private_workspace: bool = False, x_pm: Optional[torch.Tensor] = None, want_img: bool = False):
        for t, nme in ((x, "x"), (w1, "conv1.weight"), (w2, "conv2.weight")):
            _need_gpu(t, nme)
        self.x = x.contiguous()
        B, C, N = self.x.shape
        k = idx.shape[-1]
        if idx.dtype != torch.int32 or not idx.is_contiguous():
            raise RuntimeError("gfs3d.edgeconv: idx must be contiguous int32 (B,N,k)")
        if w1.numel() != EC_WIDTH * 2 * C or w2.numel() != EC_WIDTH * EC_WIDTH:
            raise NotImplementedError("gfs3d.edgeconv: only EdgeConv widths [64, 64] are implemented in HIP")
        self.B, self.C, self.N, self.k, self.idx = B, C, N, k, idx
        if x_pm is not None and (tuple(x_pm.shape) != (B, N, C) or not x_pm.is_contiguous() or x_pm.dtype != torch.float32):
            raise RuntimeError("gfs3d.edgeconv: x_pm must be the contiguous (B,N,C) fp32 point-major copy of x")
        self.x_pm = x_pm
        self.training = bool(training)
        self.compute = _compute                      # the backward of this run uses the same mode
        self.scalars = (float(eps), float(momentum), float(slope))
        self.w1, self.w2 = w1.contiguous(), w2.contiguous()
        self.g1, self.b1, self.g2, self.b2 = g1, b1, g2, b2
        self.rm = (rm1, rv1, rm2, rv2)
        dev = self.x.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.P = torch.empty((B, N, EC_WIDTH), **f32)
        self.Q = torch.empty((B, N, EC_WIDTH), **f32)
        self.stats = torch.empty((8, EC_WIDTH), **f32)
        if self.training:
            self.SP = torch.empty((B, N, EC_WIDTH), **f32)
            self.SQ = torch.empty((B, N, EC_WIDTH), **f32)
            self.indeg = torch.empty((B, N), **f32)
            self.hsel = torch.empty((B, N, EC_WIDTH), **f32)
            self.jsel = torch.empty((B, N, EC_WIDTH), dtype=torch.uint8, device=dev)
            self.Pp = torch.empty((B, N, EC_WIDTH), **f32)        # BN-folded projections, reused by the backward
            self.Qp = torch.empty((B, N, EC_WIDTH), **f32)
            # reverse kNN graph in CSR form (fixed in-edge order: no float atomics in the backward)
            self.csr_rowptr = torch.empty(B * N + 1, dtype=torch.int32, device=dev)
            self.csr_pos = torch.empty((B, N, k), dtype=torch.int32, device=dev)
        else:
            self.SP = self.SQ = self.indeg = self.hsel = self.jsel = self.Pp = self.Qp = self.csr_rowptr = self.csr_pos = None
        self.y_pm = torch.empty((B, N, EC_WIDTH), **f32)      # point-major copy: input of the point-wise stack (K4)
        self.y_sq = torch.empty(B * N + 16, **f32)            # |y_n|^2 for the next layer's kNN (+ the slack its 16-B reads need)
        self.y = torch.empty((B, EC_WIDTH, N), **f32)
        # operand rows of y for the next block's screened kNN ([hi | mid] bf16 pairs, + the zeroed slack rows its last stage reads)
        self.y_img = torch.empty((B * N + _lib.KNN_IMG_SLACK_ROWS) * EC_WIDTH, dtype=torch.int32, device=dev) if want_img else None
        nbytes = _lib.lib().gfs3d_edgeconv_workspace_bytes(B, C, N, k)
        self._private_ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev) if private_workspace else None
        self._nbytes = nbytes

    # -- argument structs -------------------------------------------------
    def _args(self, a: EdgeConvArgs, with_running: bool):
        eps, momentum, slope = self.scalars
        a.B, a.C, a.N, a.k = self.B, self.C, self.N, self.k
        a.training = 1 if self.training else 0
        a.compute = self.compute
        a.eps, a.momentum, a.slope = eps, momentum, slope
        a.x, a.idx, a.w1 = self.x.data_ptr(), self.idx.data_ptr(), self.w1.data_ptr()
        a.bn1_w, a.bn1_b = self.g1.data_ptr(), self.b1.data_ptr()
        a.w2 = self.w2.data_ptr()
        a.bn2_w, a.bn2_b = self.g2.data_ptr(), self.b2.data_ptr()
        rm1, rv1, rm2, rv2 = self.rm if with_running else (None,) * 4
        a.bn1_rm = rm1.data_ptr() if rm1 is not None else None
        a.bn1_rv = rv1.data_ptr() if rv1 is not None else None
        a.bn2_rm = rm2.data_ptr() if rm2 is not None else None
        a.bn2_rv = rv2.data_ptr() if rv2 is not None else None
        a.P, a.Q, a.stats = self.P.data_ptr(), self.Q.data_ptr(), self.stats.data_ptr()
        for name in ("SP", "SQ", "indeg", "hsel", "jsel", "y_pm", "Pp", "Qp", "csr_rowptr", "csr_pos"):
            t = getattr(self, name)
            setattr(a, name, t.data_ptr() if t is not None else None)
        a.y_cf, a.y_cf_bstride = (self.y.data_ptr() if self.y is not None else None), EC_WIDTH * self.N
        ws = self._private_ws if self._private_ws is not None else _workspace(self.x.device, self._nbytes)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.x_pm = self.x_pm.data_ptr() if self.x_pm is not None else None
        a.y_sqnorm = self.y_sq.data_ptr() if self.y_sq is not None else None
        a.y_knn_img = self.y_img.data_ptr() if (self.y_img is not None and with_running) else None     # (forward only)
        return ws

    def _bwd_args(self, dy, dy_pm=None):
        dev = self.x.device
        f32 = dict(dtype=torch.float32, device=dev)
        if dy is None and dy_pm is None:
            raise RuntimeError("gfs3d.edgeconv: backward without a gradient")
        # a channel slice of a wider (B, C', N) gradient (torch.cat of the blocks' outputs upstream) is read in place through
        # its batch stride - .contiguous() was a 17 MB copy per block and step
        if dy is not None and not (dy.dim() == 3 and dy.dtype == torch.float32 and dy.stride(2) == 1 and dy.stride(1) == self.N
                                   and dy.stride(0) >= EC_WIDTH * self.N):
            dy = dy.contiguous()
        self.dy = dy
        self.dy_pm = dy_pm.contiguous() if dy_pm is not None else None
        self.dx = torch.empty_like(self.x)
        self.dw1, self.dw2 = torch.empty_like(self.w1), torch.empty_like(self.w2)
        self.dg1, self.db1, self.dg2, self.db2 = (torch.empty(EC_WIDTH, **f32) for _ in range(4))
        ba = EdgeConvBwdArgs()
        ws = self._args(ba.f, with_running=False)
        ba.dy_cf, ba.dy_bstride = (self.dy.data_ptr() if self.dy is not None else None), (self.dy.stride(0) if self.dy is not None else EC_WIDTH * self.N)
        ba.dy_pm = self.dy_pm.data_ptr() if self.dy_pm is not None else None
        ba.dx = self.dx.data_ptr()
        ba.dw1, ba.dbn1_w, ba.dbn1_b = self.dw1.data_ptr(), self.dg1.data_ptr(), self.db1.data_ptr()
        ba.dw2, ba.dbn2_w, ba.dbn2_b = self.dw2.data_ptr(), self.dg2.data_ptr(), self.db2.data_ptr()
        return ba, ws

    # -- execution --------------------------------------------------------
    def forward_phase(self, phase: int, acc: torch.Tensor, stat_count: float):
        a = EdgeConvArgs()
        ws = self._args(a, with_running=True)
        with torch.cuda.device(self.x.device):
            rc = _lib.lib().gfs3d_edgeconv_forward_phase_f32(ctypes.byref(a), phase, _p(acc), float(stat_count), _stream())
        _lib.check(rc, "gfs3d_edgeconv_forward_phase_f32")
        del ws

    def backward_phase(self, phase: int, acc: torch.Tensor, stat_count: float, dy=None, dy_pm=None):
        if phase == 0:
            self._ba = self._bwd_args(dy, dy_pm)
        ba, ws = self._ba
        with torch.cuda.device(self.x.device):
            rc = _lib.lib().gfs3d_edgeconv_backward_phase_f32(ctypes.byref(ba), phase, _p(acc), float(stat_count), _stream())
        _lib.check(rc, "gfs3d_edgeconv_backward_phase_f32")
        if phase == 2:
            del self._ba
            return self._take_grads()
        return None

    def forward(self, sync: Optional[BNSync] = None):
        if sync is None or not self.training:
            a = EdgeConvArgs()
            ws = self._args(a, with_running=True)
            with torch.cuda.device(self.x.device):
                rc = _lib.lib().gfs3d_edgeconv_forward_f32(ctypes.byref(a), _stream())
            _lib.check(rc, "gfs3d_edgeconv_forward_f32")
            del ws
            return self.y
        acc = torch.zeros(512, dtype=torch.float64, device=self.x.device)
        count = float(self.B) * self.N * self.k * sync.world_size
        self.forward_phase(0, acc, count)
        sync.all_reduce(acc[0:128])
        self.forward_phase(1, acc, count)
        sync.all_reduce(acc[128:256])
        self.forward_phase(2, acc, count)
        return self.y

    def backward(self, dy, sync: Optional[BNSync] = None, dy_pm=None):
        if sync is None:
            ba, ws = self._bwd_args(dy, dy_pm)
            with torch.cuda.device(self.x.device):
                rc = _lib.lib().gfs3d_edgeconv_backward_f32(ctypes.byref(ba), _stream())
            _lib.check(rc, "gfs3d_edgeconv_backward_f32")
            del ws
        else:
            acc = torch.zeros(512, dtype=torch.float64, device=self.x.device)
            count = float(self.B) * self.N * self.k * sync.world_size
            self.backward_phase(0, acc, count, dy, dy_pm)
            sync.all_reduce(acc[0:128])
            self.backward_phase(1, acc, count)
            sync.all_reduce(acc[128:256])
            return self.backward_phase(2, acc, count)
        return self._take_grads()

    def _take_grads(self):
        """Hand the gradient tensors to the caller and forget them: autograd's AccumulateGrad adopts a gradient
        without a copy only if nothing else references it."""
        out = (self.dx, self.dw1, self.dg1, self.db1, self.dw2, self.dg2, self.db2)
        self.dx = self.dw1 = self.dg1 = self.db1 = self.dw2 = self.dg2 = self.db2 = self.dy = self.dy_pm = None
        return out


class EdgeConvFunction(torch.autograd.Function):
    """y = max_k LReLU(BN2(W2 LReLU(BN1(W1 [x_j - x_i ; x_i]))))  for one block
    (model/dgcnn.py:116-118), forward and backward in HIP.  Returns (y (B,64,N), y_pm (B,N,64)): the same
    values channel-first (the reference layout) and point-major (what the point-wise stack consumes);
    gradients arriving through either are added."""

    @staticmethod
    def forward(ctx, x, idx, w1, g1, b1, w2, g2, b2, rm1, rv1, rm2, rv2, training, eps, momentum, slope, sync, x_pm, want_img=False):
        run = EdgeConvRun(x, idx, w1, g1, b1, rm1, rv1, w2, g2, b2, rm2, rv2, training, eps, momentum, slope, x_pm=x_pm,
                          want_img=want_img)
        y = run.forward(sync)
        run.y = None                     # outputs must not be reachable from ctx (reference cycle through grad_fn)
        ctx.set_materialize_grads(False)
        if training:
            ctx.c = run
            ctx.sync = sync
        y_sq = run.y_sq
        run.y_sq = None
        ctx.mark_non_differentiable(y_sq)
        if want_img:
            y_img = run.y_img
            run.y_img = None
            ctx.mark_non_differentiable(y_img)
            return y, run.y_pm.detach(), y_sq, y_img
        return y, run.y_pm.detach(), y_sq      # y_pm is an alias: the backward reads it (sign of the activation)

    @staticmethod
    def backward(ctx, dy, dy_pm, _dy_sq=None, _dy_img=None):
        if not hasattr(ctx, "c"):
            raise RuntimeError("gfs3d.edgeconv: backward is implemented for training mode (batch statistics) only")
        if dy is None and dy_pm is None:
            return (None,) * 19
        grads = ctx.c.backward(dy, ctx.sync, dy_pm)
        dx, dw1, dg1, db1, dw2, dg2, db2 = grads
        return (dx, None, dw1, dg1, db1, dw2, dg2, db2) + (None,) * 11


def edgeconv_block(x, idx32, conv1_w, bn1, conv2_w, bn2, training: bool, slope: float = 0.2,
                   sync: Optional[BNSync] = None, want_pm: bool = False, x_pm: Optional[torch.Tensor] = None,
                   track: bool = True, want_sq: bool = False, want_img: bool = False):
    """Run one EdgeConv block given the two conv weights and the two
    nn.BatchNorm2d modules (their running stats are updated in place when
    training, like nn.BatchNorm2d does).  `sync`: share the batch statistics between ranks.  `x_pm`: the
    (B,N,C) point-major values of x when the caller has them (the previous block's second output) - the input
    projections then run as one MFMA GEMM; the gradient still flows through x.  `track=False`: the caller
    bumps `num_batches_tracked` itself (one multi-tensor add for a whole network)."""
    if training and track:
        for bn in (bn1, bn2):
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(1)
    mom = bn1.momentum if bn1.momentum is not None else 0.1
    out = EdgeConvFunction.apply(x, idx32, conv1_w, bn1.weight, bn1.bias, conv2_w, bn2.weight, bn2.bias,
                                 bn1.running_mean, bn1.running_var, bn2.running_mean, bn2.running_var,
                                 bool(training), float(bn1.eps), float(mom), float(slope), sync,
                                 x_pm.detach() if x_pm is not None else None, bool(want_img))
    y, y_pm, y_sq = out[0], out[1], out[2]
    if want_img:                             # (+ the next block's screen operand rows: `knn_topk(..., x_img=)`)
        return y, y_pm, y_sq, out[3]
    if want_sq:                              # (y, y_pm, |y|^2 in the kNN kernel's order: `knn_topk(..., sqnorm=)` of the next block)
        return y, y_pm, y_sq
    return (y, y_pm) if want_pm else y


# --------------------------------------------------------------------------
# K4
# --------------------------------------------------------------------------
class PointwiseRun:
    """One invocation of the point-wise stack [Conv1d 1x1 (+bias) -> BatchNorm1d -> LeakyReLU(slope_l)] x L on point-major
    segments of equal width (a multiple of 64): the DGCNN stack (model/dgcnn.py:64-80, 121-122), BaseLearner
    (dgcnn.py:130-152: ReLU between, nothing after) and the fusion layer (capl.py:64-65)."""

    def __init__(self, xs, layers, training, eps, momentum, slope, private_workspace: bool = False, biases=None, slopes=None,
                 want_cf: bool = True, want_pm: bool = False):
        if not 1 <= len(xs) <= PW_MAX_SEGS or not 1 <= len(layers) <= PW_MAX_LAYERS:
            raise NotImplementedError("gfs3d.pointwise: 1..4 input segments and 1..4 layers are implemented in HIP")
        segw = int(xs[0].shape[-1])
        self.compute = _compute
        self.training = bool(training)
        self.h_dtype = _h_dtype(self.compute, self.training)
        for t in xs:
            if t.is_cuda and t.dtype == self.h_dtype and t.dtype != torch.float32 and t.dtype == xs[0].dtype:
                pass                                   # a 16-bit input in the mode's own format (the packed fusion input)
            else:
                _need_gpu(t, "x_seg")
            if t.dim() != 3 or segw % PW_SEG_WIDTH or t.shape != xs[0].shape or t.dtype != xs[0].dtype:
                raise RuntimeError("gfs3d.pointwise: inputs must be equal-shape point-major (B,N,64 m) tensors of one type")
        if segw * len(xs) > 1024:
            raise NotImplementedError("gfs3d.pointwise: at most 1024 input channels")
        if not (want_cf or want_pm):
            raise ValueError("gfs3d.pointwise: no output requested")
        self.segw = segw
        self.xs = [t.contiguous() for t in xs]
        self.B, self.N = self.xs[0].shape[:2]
        self.layers = [(w.contiguous(), g, b, rm, rv) for (w, g, b, rm, rv) in layers]
        self.widths = [int(w.shape[0]) for (w, *_r) in self.layers]
        prev = segw * len(xs)
        for (w, *_r), c in zip(self.layers, self.widths):
            if w.numel() != c * prev or c % 64 or c > 1024:
                raise NotImplementedError("gfs3d.pointwise: layer widths must be multiples of 64 (<= 1024) and chain")
            prev = c
        L = len(self.layers)
        self.biases = list(biases) if biases is not None else [None] * L
        self.slopes = [float(v) for v in slopes] if slopes is not None else None
        if len(self.biases) != L or (self.slopes is not None and len(self.slopes) != L):
            raise ValueError("gfs3d.pointwise: one bias / slope per layer")
        self.scalars = (float(eps), float(momentum), float(slope))
        dev = self.xs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        M = self.B * self.N
        self.h = [torch.empty((M, c), dtype=self.h_dtype, device=dev) if (self.training or l < L - 1) else None
                  for l, c in enumerate(self.widths)]
        # reduced training call on fp32 inputs: the first GEMM leaves its 16-bit operands of x behind for the backward's weight gradient
        self.x16 = (torch.empty((M, segw * len(self.xs)), dtype=self.h_dtype, device=dev)
                    if (self.h_dtype != torch.float32 and self.xs[0].dtype == torch.float32) else None)
        self.stats = [torch.empty((4, c), **f32) for c in self.widths]
        self.y = torch.empty((self.B, self.widths[-1], self.N), **f32) if want_cf else None
        self.y_pm = torch.empty((self.B, self.N, self.widths[-1]), **f32) if want_pm else None
        arr = (ctypes.c_int * L)(*self.widths)
        self._nbytes = _lib.lib().gfs3d_pointwise_workspace_bytes_ex(self.B, self.N, len(xs), segw, L, arr)
        if self._nbytes == 0:
            raise RuntimeError("gfs3d.pointwise: unsupported shape")
        self._private_ws = torch.empty(self._nbytes + 256, dtype=torch.uint8, device=dev) if private_workspace else None

    def _args(self, a: PointwiseArgs):
        eps, momentum, slope = self.scalars
        a.B, a.N, a.n_seg, a.n_layers = self.B, self.N, len(self.xs), len(self.layers)
        a.seg_width = self.segw
        a.training = 1 if self.training else 0
        a.compute = self.compute
        a.store16 = 1 if self.h_dtype != torch.float32 else 0
        a.x_type = self.compute if self.xs[0].dtype != torch.float32 else 0
        a.x16 = self.x16.data_ptr() if self.x16 is not None else None
        a.eps, a.momentum, a.slope = eps, momentum, slope
        a.per_layer_slope = 0 if self.slopes is None else 1
        for s_, t in enumerate(self.xs):
            a.x_seg[s_] = t.data_ptr()
        for l, (w, g, b, rm, rv) in enumerate(self.layers):
            a.width[l] = self.widths[l]
            a.w[l], a.bn_w[l], a.bn_b[l] = w.data_ptr(), g.data_ptr(), b.data_ptr()
            a.bn_rm[l] = rm.data_ptr() if rm is not None else None
            a.bn_rv[l] = rv.data_ptr() if rv is not None else None
            a.h[l] = self.h[l].data_ptr() if self.h[l] is not None else None
            a.stats[l] = self.stats[l].data_ptr()
            a.bias[l] = self.biases[l].data_ptr() if self.biases[l] is not None else None
            if self.slopes is not None:
                a.slope_l[l] = self.slopes[l]
        a.y_cf = self.y.data_ptr() if self.y is not None else None      # not needed by the backward
        a.y_pm = self.y_pm.data_ptr() if self.y_pm is not None else None
        ws = self._private_ws if self._private_ws is not None else _workspace(self.xs[0].device, self._nbytes)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        return ws

    def _acc(self):
        """(acc buffer, doubles per row): rows [l][2][maxw] of global sums + the library's copies of this rank's sums."""
        row = 2 * max(self.widths)
        return torch.zeros(2 * len(self.layers) * row, dtype=torch.float64, device=self.xs[0].device), row

    def forward_phase(self, phase: int, acc: torch.Tensor, stat_count: float):
        """One phase of the forward (tests / custom exchanges); needs private_workspace=True if other calls run between phases."""
        a = PointwiseArgs()
        ws = self._args(a)
        with torch.cuda.device(self.xs[0].device):
            rc = _lib.lib().gfs3d_pointwise_forward_phase_f32(ctypes.byref(a), phase, _p(acc), float(stat_count), _stream())
        _lib.check(rc, "gfs3d_pointwise_forward_phase_f32")
        del ws

    def backward_phase(self, phase: int, acc: torch.Tensor, stat_count: float, dy=None, dy_pm=None):
        if phase == 0:
            self._ba = self._bwd_args(dy, dy_pm)
        ba, ws = self._ba
        with torch.cuda.device(self.xs[0].device):
            rc = _lib.lib().gfs3d_pointwise_backward_phase_f32(ctypes.byref(ba), phase, _p(acc), float(stat_count), _stream())
        _lib.check(rc, "gfs3d_pointwise_backward_phase_f32")
        if phase == len(self.layers):
            del self._ba
            out = (self.dxs, self.dws, self.dgs, self.dbs)
            self.dxs = self.dws = self.dgs = self.dbs = self.dy = self.dy_pm = None
            return out
        return None

    def forward(self, sync: Optional["BNSync"] = None):
        a = PointwiseArgs()
        ws = self._args(a)
        L = _lib.lib()
        with torch.cuda.device(self.xs[0].device):
            if sync is None or not self.training:
                rc = L.gfs3d_pointwise_forward_f32(ctypes.byref(a), _stream())
                _lib.check(rc, "gfs3d_pointwise_forward_f32")
            else:       # synchronised BatchNorm: the column sums of every layer are all-reduced between the phases
                acc, row = self._acc()
                count = float(self.B) * self.N * sync.world_size
                for ph in range(len(self.layers) + 1):
                    rc = L.gfs3d_pointwise_forward_phase_f32(ctypes.byref(a), ph, _p(acc), count, _stream())
                    _lib.check(rc, "gfs3d_pointwise_forward_phase_f32")
                    if ph < len(self.layers):
                        sync.all_reduce(acc[ph * row:(ph + 1) * row])
        del ws
        return self.y if self.y is not None else self.y_pm

    def _bwd_args(self, dy, dy_pm=None):
        dev = self.xs[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        if dy is None and dy_pm is None:
            raise RuntimeError("gfs3d.pointwise: backward without a gradient")
        self.dy = dy.contiguous() if dy is not None else None
        self.dy_pm = dy_pm.contiguous() if dy_pm is not None else None
        x16 = self.xs[0].dtype != torch.float32            # 16-bit inputs (internal to a fused operator): their gradients are bf16
        self.dxs = [torch.empty(t.shape, dtype=torch.bfloat16 if x16 else torch.float32, device=dev) for t in self.xs]
        self.dws = [torch.empty_like(w) for (w, *_r) in self.layers]
        self.dgs = [torch.empty(c, **f32) for c in self.widths]
        self.dbs = [torch.empty(c, **f32) for c in self.widths]
        ba = PointwiseBwdArgs()
        saved = self.layers
        self.layers = [(w, g, b, None, None) for (w, g, b, rm, rv) in saved]      # running stats are not touched
        ws = self._args(ba.f)
        self.layers = saved
        ba.dy_cf = self.dy.data_ptr() if self.dy is not None else None
        ba.dy_pm = self.dy_pm.data_ptr() if self.dy_pm is not None else None
        ba.dx_type = 1 if x16 else 0
        for s_, t in enumerate(self.dxs):
            ba.dx_seg[s_] = t.data_ptr()
        for l in range(len(self.layers)):
            ba.dw[l], ba.dbn_w[l], ba.dbn_b[l] = self.dws[l].data_ptr(), self.dgs[l].data_ptr(), self.dbs[l].data_ptr()
        return ba, ws

    def backward(self, dy, sync: Optional["BNSync"] = None, dy_pm=None):
        dev = self.xs[0].device
        ba, ws = self._bwd_args(dy, dy_pm)
        with torch.cuda.device(dev):
            if sync is None:
                rc = _lib.lib().gfs3d_pointwise_backward_f32(ctypes.byref(ba), _stream())
                _lib.check(rc, "gfs3d_pointwise_backward_f32")
            else:
                acc, row = self._acc()
                count = float(self.B) * self.N * sync.world_size
                nl = len(self.layers)
                for ph in range(nl + 1):
                    rc = _lib.lib().gfs3d_pointwise_backward_phase_f32(ctypes.byref(ba), ph, _p(acc), count, _stream())
                    _lib.check(rc, "gfs3d_pointwise_backward_phase_f32")
                    if ph < nl:                       # phase ph leaves the sums of layer nl - 1 - ph
                        l = nl - 1 - ph
                        sync.all_reduce(acc[l * row:(l + 1) * row])
        del ws
        out = (self.dxs, self.dws, self.dgs, self.dbs)
        self.dxs = self.dws = self.dgs = self.dbs = self.dy = self.dy_pm = None          # see EdgeConvRun._take_grads
        return out


class PointwiseStackFunction(torch.autograd.Function):
    """act(bn_L(... act(bn_1(W_1 cat(xs))) ...)); tensors = xs + [w, gamma, beta] * L + [rm, rv] * L + [bias] * L (bias
    entries may be None).  opts = (slopes or None, want_cf, want_pm[, pack]); returns y_cf (B, C_L, N), y_pm (B, N, C_L) or both.
    pack = (channel_first flags, ld): the n_seg tensors are SOURCES in mixed layouts (channel-first (B, w, N) / point-major
    (B, N, w)) that are first concatenated into one zero-padded point-major (B, N, ld) matrix (the torch.cat of capl.py:357-359)
    INSIDE the operator - the packed matrix and its gradient are then internal tensors: 16-bit in the reduced modes."""

    @staticmethod
    def forward(ctx, n_seg, n_layers, training, eps, momentum, slope, sync, opts, *tensors):
        slopes, want_cf, want_pm = opts[:3]
        pack = opts[3] if len(opts) > 3 else None
        xs = tensors[:n_seg]
        ctx.pack = None
        if pack is not None:
            cf_flags, ld = pack
            srcs = []
            for t in xs:
                _need_gpu(t, "src")
                srcs.append(t.contiguous())
            B = srcs[0].shape[0]
            N = srcs[0].shape[2] if cf_flags[0] else srcs[0].shape[1]
            for t, cf in zip(srcs, cf_flags):
                if t.dim() != 3 or t.shape[0] != B or (t.shape[2] if cf else t.shape[1]) != N:
                    raise RuntimeError("gfs3d.pointwise_stack_packed: sources must be (B, w, N) channel-first or (B, N, w) point-major")
            packed = torch.empty((B, N, int(ld)), dtype=_h_dtype(_compute, bool(training)), device=srcs[0].device)
            pa = _pack_args(B, N, srcs, cf_flags, packed)
            with torch.cuda.device(packed.device):
                rc = _lib.lib().gfs3d_pack_rows_f32(ctypes.byref(pa), _stream())
            _lib.check(rc, "gfs3d_pack_rows_f32")
            ctx.pack = (B, N, tuple(cf_flags), [tuple(t.shape) for t in srcs])
            xs = (packed,)
        par = tensors[n_seg:n_seg + 3 * n_layers]
        run_ = tensors[n_seg + 3 * n_layers:n_seg + 5 * n_layers]
        biases = tensors[n_seg + 5 * n_layers:]
        layers = [(par[3 * l], par[3 * l + 1], par[3 * l + 2], run_[2 * l], run_[2 * l + 1]) for l in range(n_layers)]
        run = PointwiseRun(xs, layers, training, eps, momentum, slope, biases=list(biases) if biases else None, slopes=slopes,
                           want_cf=want_cf, want_pm=want_pm)
        run.forward(sync)
        y, y_pm = run.y, run.y_pm
        run.y = run.y_pm = None
        if training:
            ctx.c = run
            ctx.sync = sync
        ctx.n = (n_seg, n_layers, want_cf, want_pm, [b is not None for b in biases] if biases else [False] * n_layers)
        if ctx.pack is not None:
            ctx.pack_needs = tuple(ctx.needs_input_grad[8:8 + n_seg])
        if want_cf and want_pm:
            return y, y_pm
        return y if want_cf else y_pm

    @staticmethod
    def backward(ctx, *douts):
        if not hasattr(ctx, "c"):
            raise RuntimeError("gfs3d.pointwise: backward is implemented for training mode (batch statistics) only")
        n_seg, n_layers, want_cf, want_pm, has_bias = ctx.n
        if want_cf and want_pm:
            dy, dy_pm = douts
        else:
            dy, dy_pm = (douts[0], None) if want_cf else (None, douts[0])
        dxs, dws, dgs, dbs = ctx.c.backward(dy, ctx.sync, dy_pm=dy_pm)
        if ctx.pack is not None:            # the adjoint of the pack: columns of the (bf16 / fp32) packed gradient back to the sources
            B, N, cf_flags, shapes = ctx.pack
            dpacked = dxs[0]
            outs = [torch.empty(sh, dtype=torch.float32, device=dpacked.device) if need else None for sh, need in zip(shapes, ctx.pack_needs)]
            # sources that need no gradient still occupy their columns: give them scratch so that offsets stay right
            tmp = [o if o is not None else torch.empty(sh, dtype=torch.float32, device=dpacked.device) for o, sh in zip(outs, shapes)]
            pa = _pack_args(B, N, tmp, cf_flags, dpacked)
            with torch.cuda.device(dpacked.device):
                rc = _lib.lib().gfs3d_unpack_rows_f32(ctypes.byref(pa), _stream())
            _lib.check(rc, "gfs3d_unpack_rows_f32")
            dxs = outs
        par = []
        for l in range(n_layers):
            par += [dws[l], dgs[l], dbs[l]]
        # a bias in front of a training-mode BatchNorm has an exactly zero gradient (the batch mean absorbs it)
        dbias = tuple(torch.zeros_like(dgs[l]) if has_bias[l] else None for l in range(n_layers))
        return (None,) * 8 + tuple(dxs) + tuple(par) + (None,) * (2 * n_layers) + dbias


def pointwise_stack(xs_pm, parts, training: bool, slope: float = 0.2, track: bool = True, sync: Optional[BNSync] = None,
                    slopes=None, want_cf: bool = True, want_pm: bool = False):
    """Run a conv1d stack given [(nn.Conv1d, nn.BatchNorm1d)] on point-major (B,N,64 m) segments.  `sync`: share the batch
    statistics between the ranks of a block-sharded batch (phased C ABI, one all-reduce per layer and pass).  `slopes`: the
    activation after each layer as a LeakyReLU slope (0 = ReLU, 1 = none), default `slope` everywhere.  Convolution biases
    are honoured.  Returns the channel-first result, the point-major one, or (cf, pm)."""
    if training and track:                      # nn.BatchNorm bookkeeping of the stack in one multi-tensor add
        counters = [bn.num_batches_tracked for _c, bn in parts if bn.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)
    bn0 = parts[0][1]
    mom = bn0.momentum if bn0.momentum is not None else 0.1
    par, run_, bias = [], [], []
    for conv, bn in parts:                      # conv: an nn.Conv1d or a (weight, bias) pair (e.g. a K-padded weight)
        w_, b_ = conv if isinstance(conv, tuple) else (conv.weight, conv.bias)
        par += [w_, bn.weight, bn.bias]
        run_ += [bn.running_mean, bn.running_var]
        bias.append(b_)
    opts = (tuple(float(v) for v in slopes) if slopes is not None else None, bool(want_cf), bool(want_pm))
    return PointwiseStackFunction.apply(len(xs_pm), len(parts), bool(training), float(bn0.eps), float(mom), float(slope), sync, opts,
                                        *xs_pm, *par, *run_, *bias)


def pointwise_stack_packed(sources, cf_flags, ld: int, parts, training: bool, slope: float = 0.2, track: bool = True,
                           sync: Optional[BNSync] = None, slopes=None, want_cf: bool = True, want_pm: bool = False):
    """`pointwise_stack([pack_rows(sources, cf_flags, ld)], parts, ...)` as ONE operator: the concatenated matrix (the fusion
    layer's input, capl.py:357-359) and its gradient never cross an autograd boundary, so in the reduced modes they are kept
    in 16 bits (the forward GEMM copies the rows to LDS as loaded; 100 instead of 201 MB each way at configs[4] size)."""
    if training and track:
        counters = [bn.num_batches_tracked for _c, bn in parts if bn.num_batches_tracked is not None]
        if counters:
            torch._foreach_add_(counters, 1)
    bn0 = parts[0][1]
    mom = bn0.momentum if bn0.momentum is not None else 0.1
    par, run_, bias = [], [], []
    for conv, bn in parts:
        w_, b_ = conv if isinstance(conv, tuple) else (conv.weight, conv.bias)
        par += [w_, bn.weight, bn.bias]
        run_ += [bn.running_mean, bn.running_var]
        bias.append(b_)
    opts = (tuple(float(v) for v in slopes) if slopes is not None else None, bool(want_cf), bool(want_pm),
            (tuple(bool(c) for c in cf_flags), int(ld)))
    return PointwiseStackFunction.apply(len(sources), len(parts), bool(training), float(bn0.eps), float(mom), float(slope), sync, opts,
                                        *sources, *par, *run_, *bias)


class WordsFusionFunction(torch.autograd.Function):
    """The geometric-word branch and the fusion layer of `getFeatures` as ONE operator (model/capl.py:344-359, 64-65):
        soft = softmax(scale cos(word, edge feature));  word = argmax;  y = act(bn(W [soft | others] + b))
    Forward: word assignment (K3) -> pack -> fusion stack (K4).  Backward: the stack's input gradient stays packed; its other
    sources are unpacked, its soft columns go through `gfs3d_word_soft_backward_rows` (d = soft (dsoft - <soft, dsoft>)) and then
    the word-gradient kernel in its plain logits form - soft and its gradient are read once each, the channel-first dsoft is
    never written, and `soft`, the packed matrix and its gradient never cross an autograd boundary (16-bit in the reduced modes).
    tensors = [feat (B,C,N), words (Kw,C)] + others + [w, gamma, beta, rm, rv, bias]; returns (y_cf (B,Cout,N), word (B,N) int32)."""

    @staticmethod
    def forward(ctx, scale, other_cf, ld, training, eps, momentum, slope, sync, n_other, *tensors):
        feat, words = tensors[0], tensors[1]
        others = tensors[2:2 + n_other]
        w, gamma, beta, rm, rv, bias = tensors[2 + n_other:]
        _need_gpu(feat, "feat")
        _need_gpu(words, "words")
        feat = feat.contiguous()
        B, C, N = feat.shape
        Kw = words.shape[0]
        soft, word, _ = word_assign_raw(feat, words, scale, want_onehot=False)
        srcs = [soft] + [t.contiguous() for t in others]
        flags = (True,) + tuple(bool(c) for c in other_cf)
        for t, cf in zip(srcs, flags):
            _need_gpu(t, "src")
            if t.dim() != 3 or t.shape[0] != B or (t.shape[2] if cf else t.shape[1]) != N:
                raise RuntimeError("gfs3d.words_fusion: sources must be (B, w, N) channel-first or (B, N, w) point-major")
        packed = torch.empty((B, N, int(ld)), dtype=_h_dtype(_compute, bool(training)), device=feat.device)
        pa = _pack_args(B, N, srcs, flags, packed)
        with torch.cuda.device(packed.device):
            rc = _lib.lib().gfs3d_pack_rows_f32(ctypes.byref(pa), _stream())
        _lib.check(rc, "gfs3d_pack_rows_f32")
        run = PointwiseRun([packed], [(w, gamma, beta, rm, rv)], training, eps, momentum, slope, biases=[bias])
        run.forward(sync)
        y = run.y
        run.y = None
        ctx.mark_non_differentiable(word)
        if training:
            ctx.c, ctx.sync = run, sync
            ctx.save_for_backward(feat, words, soft)
            ctx.meta = (float(scale), flags, [tuple(t.shape) for t in srcs], _compute, bias is not None, n_other)
        return y, word

    @staticmethod
    def backward(ctx, dy, _dword):
        if not hasattr(ctx, "c"):
            raise RuntimeError("gfs3d.words_fusion: backward is implemented for training mode (batch statistics) only")
        feat, words, soft = ctx.saved_tensors
        scale, flags, shapes, compute, has_bias, n_other = ctx.meta
        dxs, dws, dgs, dbs = ctx.c.backward(dy, ctx.sync)
        dpacked = dxs[0]
        B, Kw, N = soft.shape
        dev = dpacked.device
        need = ctx.needs_input_grad[9:]
        outs = [None] + [torch.empty(sh, dtype=torch.float32, device=dev) if need[2 + i] else None for i, sh in enumerate(shapes[1:])]
        widths = [sh[1] if cf else sh[2] for sh, cf in zip(shapes, flags)]
        pa = _pack_args(B, N, outs, flags, dpacked, widths=widths)
        L = _lib.lib()
        dfeat = None
        with torch.cuda.device(dev):
            if any(o is not None for o in outs):
                _lib.check(L.gfs3d_unpack_rows_f32(ctypes.byref(pa), _stream()), "gfs3d_unpack_rows_f32")
            if need[0]:
                d = torch.empty_like(soft)
                rc = L.gfs3d_word_soft_backward_rows(_p(dpacked), _lib.COMPUTE_BF16 if dpacked.dtype == torch.bfloat16 else 0,
                                                     int(dpacked.shape[-1]), _p(soft), B, Kw, N, _p(d), _stream())
                _lib.check(rc, "gfs3d_word_soft_backward_rows")
                dfeat = proto_backward_feat(feat, words, scale, d, None, compute=compute)
        dbias = torch.zeros_like(dgs[0]) if has_bias else None      # a bias in front of a training-mode BatchNorm: exactly zero
        return (None,) * 9 + (dfeat, None) + tuple(outs[1:]) + (dws[0], dgs[0], dbs[0], None, None, dbias)


def words_fusion(feat, words, scale, others, other_cf, ld: int, conv_w, conv_b, bn, training: bool, slope: float = 0.2, track: bool = True,
                 sync: Optional[BNSync] = None):
    """-> (point_feat (B, Cout, N), word (B, N) int32): `pointwise_stack_packed([softmax words | others], ...)` with the word
    assignment inside (see WordsFusionFunction).  `conv_w` (Cout, ld): the fusion convolution's weight, K-padded to `ld`."""
    if training and track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = bn.momentum if bn.momentum is not None else 0.1
    return WordsFusionFunction.apply(float(scale), tuple(bool(c) for c in other_cf), int(ld), bool(training), float(bn.eps), float(mom),
                                     float(slope), sync, len(others), feat, words, *others, conv_w, bn.weight, bn.bias,
                                     bn.running_mean, bn.running_var, conv_b)


class LinearFunction(torch.autograd.Function):
    """y = x W^T for a bias-free 1x1 convolution without BatchNorm on a point-major input: x_pm (B, N, K), w (64 n, K) ->
    n channel-first (B, 64, N) tensors (the q / k / v projections of model/attention.py:39-41 as one GEMM)."""

    @staticmethod
    def forward(ctx, x_pm, w):
        _need_gpu(x_pm, "x_pm")
        x_pm, w2 = x_pm.contiguous(), w.reshape(w.shape[0], -1).contiguous()
        B, N, K = x_pm.shape
        Nc = w2.shape[0]
        L = _lib.lib()
        nbytes = L.gfs3d_linear_workspace_bytes(B, N, K, Nc)
        if nbytes == 0 or w2.shape[1] != K:
            raise NotImplementedError("gfs3d.linear: K a multiple of 64 (<= 1024) and 64, 128, 192 or 256 outputs are implemented")
        outs = [torch.empty((B, 64, N), dtype=torch.float32, device=x_pm.device) for _ in range(Nc // 64)]
        ptrs = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        ws = _workspace(x_pm.device, nbytes)
        with torch.cuda.device(x_pm.device):
            rc = L.gfs3d_linear_forward_f32(_p(x_pm), B, N, K, _p(w2), Nc, ptrs, _compute, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "gfs3d_linear_forward_f32")
        ctx.save_for_backward(x_pm, w2)
        ctx.compute = _compute
        ctx.wshape = w.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        x_pm, w2 = ctx.saved_tensors
        B, N, K = x_pm.shape
        Nc = w2.shape[0]
        ds = [d.contiguous() if d is not None else torch.zeros((B, 64, N), dtype=torch.float32, device=x_pm.device) for d in douts]
        ptrs = (ctypes.c_void_p * len(ds))(*[t.data_ptr() for t in ds])
        dx, dw = torch.empty_like(x_pm), torch.empty_like(w2)
        L = _lib.lib()
        nbytes = L.gfs3d_linear_workspace_bytes(B, N, K, Nc)
        ws = _workspace(x_pm.device, nbytes)
        with torch.cuda.device(x_pm.device):
            rc = L.gfs3d_linear_backward_f32(_p(x_pm), B, N, K, _p(w2), Nc, ptrs, _p(dx), _p(dw), ctx.compute, _p(ws), ws.numel(), _stream())
        _lib.check(rc, "gfs3d_linear_backward_f32")
        return dx, dw.reshape(ctx.wshape)


def linear_cf(x_pm: torch.Tensor, w: torch.Tensor):
    """x_pm (B, N, K) point-major, w (64 n, K[, 1]) -> tuple of n channel-first (B, 64, N) tensors."""
    return LinearFunction.apply(x_pm, w)


def _pack_args(B, N, tensors, cf_flags, dst, widths=None):
    """`tensors[s]` may be None in an unpack call (the source's columns are skipped); `widths` then carries its width."""
    a = _lib.PackArgs()
    a.B, a.N, a.n_src = B, N, len(tensors)
    for s_, (t, cf) in enumerate(zip(tensors, cf_flags)):
        a.src[s_] = t.data_ptr() if t is not None else None
        a.width[s_] = int(widths[s_]) if widths is not None else int(t.shape[1] if cf else t.shape[2])
        a.channel_first[s_] = 1 if cf else 0
    a.dst, a.dst_ld = dst.data_ptr(), int(dst.shape[-1])
    a.dst_type = {torch.float32: 0, torch.bfloat16: _lib.COMPUTE_BF16, torch.float16: _lib.COMPUTE_F16}[dst.dtype]
    return a


class PackRowsFunction(torch.autograd.Function):
    """Concatenate along the channels into one point-major (B, N, ld) matrix; sources are channel-first (B, w, N) or
    point-major (B, N, w); columns past the total width are zero (the torch.cat of model/capl.py:357-359 in the layout the
    point-major GEMMs read)."""

    @staticmethod
    def forward(ctx, ld, cf_flags, *tensors):
        if not 1 <= len(tensors) <= _lib.PACK_MAX_SRC:
            raise NotImplementedError("gfs3d.pack_rows: 1..8 sources")
        ts = []
        for t in tensors:
            _need_gpu(t, "src")
            ts.append(t.contiguous())
        B = ts[0].shape[0]
        N = ts[0].shape[2] if cf_flags[0] else ts[0].shape[1]
        for t, cf in zip(ts, cf_flags):
            if t.dim() != 3 or t.shape[0] != B or (t.shape[2] if cf else t.shape[1]) != N:
                raise RuntimeError("gfs3d.pack_rows: sources must be (B, w, N) channel-first or (B, N, w) point-major")
        dst = torch.empty((B, N, int(ld)), dtype=torch.float32, device=ts[0].device)
        a = _pack_args(B, N, ts, cf_flags, dst)
        with torch.cuda.device(dst.device):
            rc = _lib.lib().gfs3d_pack_rows_f32(ctypes.byref(a), _stream())
        _lib.check(rc, "gfs3d_pack_rows_f32")
        ctx.meta = (B, N, tuple(cf_flags), [tuple(t.shape) for t in ts])
        return dst

    @staticmethod
    def backward(ctx, ddst):
        B, N, cf_flags, shapes = ctx.meta
        ddst = ddst.contiguous()
        outs = [torch.empty(sh, dtype=torch.float32, device=ddst.device) if need else None
                for sh, need in zip(shapes, ctx.needs_input_grad[2:])]
        # sources that need no gradient still occupy their columns: give them scratch so that offsets stay right
        tmp = [o if o is not None else torch.empty(sh, dtype=torch.float32, device=ddst.device) for o, sh in zip(outs, shapes)]
        a = _pack_args(B, N, tmp, cf_flags, ddst)
        with torch.cuda.device(ddst.device):
            rc = _lib.lib().gfs3d_unpack_rows_f32(ctypes.byref(a), _stream())
        _lib.check(rc, "gfs3d_unpack_rows_f32")
        return (None, None) + tuple(outs)


def pack_rows(tensors, cf_flags, ld: Optional[int] = None) -> torch.Tensor:
    total = sum(int(t.shape[1] if cf else t.shape[2]) for t, cf in zip(tensors, cf_flags))
    return PackRowsFunction.apply(int(ld if ld is not None else total), tuple(bool(c) for c in cf_flags), *tensors)


# --------------------------------------------------------------------------
# K5
# --------------------------------------------------------------------------
class AttentionFunction(torch.autograd.Function):
    """dropout(softmax(q^T k * scale)) v^T -> (B, 64, N) without the (B, N, N) score matrix (model/attention.py:44-48)."""

    @staticmethod
    def forward(ctx, q, k, v, scale, p_drop, seed, keep_mask):
        for t_, nme in ((q, "q"), (k, "k"), (v, "v")):
            _need_gpu(t_, nme)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, D, N = q.shape
        if k.shape != q.shape or v.shape != q.shape:
            raise RuntimeError("gfs3d.attention: q, k, v must have the same (B, 64, N) shape")
        if keep_mask is not None:
            keep_mask = keep_mask.to(torch.uint8).contiguous()
            if tuple(keep_mask.shape) != (B, N, N):
                raise RuntimeError("gfs3d.attention: keep_mask must be (B, N, N)")
        out = torch.empty_like(q)
        lse = torch.empty((B, N), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = _lib.lib().gfs3d_attention_forward_mixed(_p(q), _p(k), _p(v), B, D, N, float(scale), float(p_drop), int(seed),
                                                          _p(keep_mask), _p(out), _p(lse), _compute, _stream())
        _lib.check(rc, "gfs3d_attention_forward_mixed")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.mask = keep_mask
        ctx.cfg = (float(scale), float(p_drop), int(seed))
        ctx.compute = _compute
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        scale, p_drop, seed = ctx.cfg
        B, D, N = q.shape
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        ws = torch.empty((B, N), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = _lib.lib().gfs3d_attention_backward_mixed(_p(q), _p(k), _p(v), _p(out), _p(lse), _p(dout), B, D, N, scale, p_drop, seed,
                                                           _p(ctx.mask), _p(dq), _p(dk), _p(dv), _p(ws), ctx.compute, _stream())
        _lib.check(rc, "gfs3d_attention_backward_mixed")
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, scale: float, p_drop: float = 0.0, seed: int = 0, keep_mask: Optional[torch.Tensor] = None):
    """Fused scaled-dot-product attention over points; `p_drop`: dropout on the attention weights (training)."""
    return AttentionFunction.apply(q, k, v, scale, p_drop, seed, keep_mask)


def attention_dropout_mask(B: int, N: int, p_drop: float, seed: int, device) -> torch.Tensor:
    """The (B, N, N) keep decisions AttentionFunction makes for (seed, p_drop)."""
    mask = torch.empty((B, N, N), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        rc = _lib.lib().gfs3d_attention_dropout_mask(B, N, float(p_drop), int(seed), _p(mask), _stream())
    _lib.check(rc, "gfs3d_attention_dropout_mask")
    return mask


# --------------------------------------------------------------------------
# K3
# --------------------------------------------------------------------------
def _bstride(t: torch.Tensor) -> int:
    """feat may be a channel slice of a bigger (B,Ctot,N) buffer: need unit
    stride over N, N stride over C; batch stride is free."""
    B, C, N = t.shape
    if t.stride(2) != 1 or (C > 1 and t.stride(1) != N):
        raise RuntimeError("gfs3d: feature tensor must be (B,C,N) with contiguous (C,N) planes")
    return t.stride(0) if B > 1 else C * N


def word_assign_raw(feat: torch.Tensor, words: torch.Tensor, scale: float = 10.0,
                    want_soft: bool = True, want_onehot: bool = True, compute: Optional[int] = None):
    """softmax(scale*cos(word, point)) over words, arg-max word and its one-hot
    (model/capl.py:344-353).  No autograd here - see WordAssignFunction."""
    _need_gpu(feat, "feat")
    _need_gpu(words, "words")
    B, C, N = feat.shape
    Kw = words.shape[0]
    words = words.contiguous()
    dev = feat.device
    soft = torch.empty((B, Kw, N), dtype=torch.float32, device=dev) if want_soft else None
    onehot = torch.empty((B, Kw, N), dtype=torch.float32, device=dev) if want_onehot else None
    assign = torch.empty((B, N), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().gfs3d_word_assign_mixed(_p(feat), _bstride(feat), B, C, N, _p(words), Kw, float(scale),
                                                _p(soft), _p(assign), _p(onehot), _compute if compute is None else compute, _stream())
    _lib.check(rc, "gfs3d_word_assign_mixed")
    return soft, assign, onehot


def proto_cosine_raw(feat: torch.Tensor, proto: torch.Tensor, scale: float = 10.0) -> torch.Tensor:
    """scale * cosine(point feature, prototype): proto (K,C) or (B,K,C) -> (B,K,N)
    (get_pred, model/capl.py:290-322)."""
    _need_gpu(feat, "feat")
    _need_gpu(proto, "proto")
    B, C, N = feat.shape
    proto = proto.contiguous()
    P = 1 if proto.dim() == 2 else proto.shape[0]
    K = proto.shape[-2]
    out = torch.empty((B, K, N), dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        rc = _lib.lib().gfs3d_proto_cosine_f32(_p(feat), _bstride(feat), B, C, N, _p(proto), P, K, float(scale),
                                               _p(out), _stream())
    _lib.check(rc, "gfs3d_proto_cosine_f32")
    return out


def proto_backward_feat(feat: torch.Tensor, proto: torch.Tensor, scale: float, dup: torch.Tensor,
                        soft: Optional[torch.Tensor] = None, compute: Optional[int] = None) -> torch.Tensor:
    """Gradient of a cosine head w.r.t. the point features: `dup` is the gradient of the logits (get_pred) or, with
    `soft` given, of the softmax over the words (word assignment).  One fused HIP kernel (softmax backward,
    (C x K).(K x points) contraction on MFMA, normalisation backward)."""
    B, C, N = feat.shape
    proto = proto.contiguous()
    P = 1 if proto.dim() == 2 else proto.shape[0]
    K = proto.shape[-2]
    dup = dup.contiguous()
    dfeat = torch.empty((B, C, N), dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        L = _lib.lib()
        nbytes = L.gfs3d_proto_backward_feat_workspace_bytes(C, K) if P == 1 else 0     # shared words: a per-call piece image
        ws = _workspace(feat.device, nbytes) if nbytes else None
        rc = L.gfs3d_proto_backward_feat_ws(_p(feat), _bstride(feat), B, C, N, _p(proto), P, K, float(scale),
                                            _p(soft), _p(dup), _p(dfeat), _compute if compute is None else compute,
                                            _p(ws), ws.numel() if ws is not None else 0, _stream())
    _lib.check(rc, "gfs3d_proto_backward_feat_ws")
    return dfeat


def proto_backward_proto(feat: torch.Tensor, proto: torch.Tensor, scale: float, dout: torch.Tensor) -> torch.Tensor:
    """Gradient of get_pred's logits w.r.t. the prototypes (shared (K,C) or per-sample (B,K,C)); HIP, bit-reproducible."""
    B, C, N = feat.shape
    proto = proto.contiguous()
    P = 1 if proto.dim() == 2 else proto.shape[0]
    K = proto.shape[-2]
    dout = dout.contiguous()
    dproto = torch.empty_like(proto)
    with torch.cuda.device(feat.device):
        L = _lib.lib()
        ws = _workspace(feat.device, L.gfs3d_proto_backward_proto_workspace_bytes(B, C, N))
        rc = L.gfs3d_proto_backward_proto_f32(_p(feat), _bstride(feat), B, C, N, _p(proto), P, K, float(scale), _p(dout), _p(dproto),
                                              _p(ws), ws.numel(), _stream())
    _lib.check(rc, "gfs3d_proto_backward_proto_f32")
    return dproto


class WordAssignFunction(torch.autograd.Function):
    """Differentiable wrt the point features (the words `gp` carry no grad,
    capl.py:49-50).  Backward: one fused HIP kernel (proto_backward_feat)."""

    @staticmethod
    def forward(ctx, feat, words, scale):
        soft, assign, onehot = word_assign_raw(feat, words, scale)
        ctx.scale = scale
        ctx.compute = _compute
        ctx.save_for_backward(feat, words, soft)
        ctx.mark_non_differentiable(onehot, assign)
        return soft, onehot, assign

    @staticmethod
    def backward(ctx, dsoft, _d1, _d2):
        feat, words, soft = ctx.saved_tensors
        if feat.shape[1] > 192:
            raise NotImplementedError("gfs3d.word_assign: the backward kernel handles up to 192 feature channels")
        return proto_backward_feat(feat, words, ctx.scale, dsoft, soft, compute=ctx.compute), None, None


class WordSoftAssignFunction(torch.autograd.Function):
    """The same without the one-hot tensor: (soft (B,Kw,N), arg-max word (B,N) int32).  For callers that look the coding up
    by word index (capl.forward) or do not use the words at all (the training step): 4 B Kw N bytes less to write, and no
    arg-max pass over a one-hot tensor afterwards."""

    @staticmethod
    def forward(ctx, feat, words, scale):
        soft, assign, _ = word_assign_raw(feat, words, scale, want_onehot=False)
        ctx.scale = scale
        ctx.compute = _compute
        ctx.save_for_backward(feat, words, soft)
        ctx.mark_non_differentiable(assign)
        return soft, assign

    @staticmethod
    def backward(ctx, dsoft, _d1):
        feat, words, soft = ctx.saved_tensors
        if feat.shape[1] > 192:
            raise NotImplementedError("gfs3d.word_assign: the backward kernel handles up to 192 feature channels")
        return proto_backward_feat(feat, words, ctx.scale, dsoft, soft, compute=ctx.compute), None, None


class ProtoCosineFunction(torch.autograd.Function):
    """get_pred with gradients to both the point features and the prototypes."""

    @staticmethod
    def forward(ctx, feat, proto, scale):
        out = proto_cosine_raw(feat, proto, scale)
        ctx.scale = scale
        ctx.save_for_backward(feat, proto)
        return out

    @staticmethod
    def backward(ctx, dout):
        feat, proto = ctx.saved_tensors
        s = ctx.scale
        dfeat = dproto = None
        if ctx.needs_input_grad[0]:
            if feat.shape[1] > 192:
                raise NotImplementedError("gfs3d.proto_cosine: the backward kernel handles up to 192 feature channels")
            dfeat = proto_backward_feat(feat, proto, s, dout)                    # fused HIP kernel
        if ctx.needs_input_grad[1]:                                              # (K, C) / (B, K, C): a skinny reduction over the points
            if feat.shape[1] > 192:
                raise NotImplementedError("gfs3d.proto_cosine: the prototype gradient handles up to 192 feature channels")
            K = proto.shape[-2]
            if K <= 32:
                dproto = proto_backward_proto(feat, proto, s, dout)              # MFMA partials + fixed-order reduction
            else:
                # the gradient of 10 cos(f, p_k) w.r.t. p_k involves no other prototype: slices of <= 32 prototypes through the
                # same kernel (the forward accepts up to 256 prototypes; ADVICE r2: this used to raise at backward time)
                parts = [proto_backward_proto(feat, proto[..., k0:k0 + 32, :], s, dout[:, k0:k0 + 32])
                         for k0 in range(0, K, 32)]
                dproto = torch.cat(parts, dim=-2)
        return dfeat, dproto, None


# --------------------------------------------------------------------------
# K6: the fused prototype head of the CAPL training step (csrc/head.hip)
# --------------------------------------------------------------------------
HEAD_MAX_FAKE = _lib.HEAD_MAX_FAKE
HEAD_C = 128


def capl_head_supported(feat: torch.Tensor, classes: int, n_fake: int) -> bool:
    """Shapes gfs3d_capl_head_* is built for: main_dim 128, at most 23 classes + background, fake-novel classes that fit
    the 32-row tile beside them."""
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 3 and feat.shape[1] == HEAD_C and 1 <= classes <= 23
            and n_fake <= HEAD_MAX_FAKE and classes + 1 + n_fake <= 32 and feat.shape[2] < (1 << 22))


def _dp_world() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class CaplTrainHeadFunction(torch.autograd.Function):
    """(point_feat (B,128,N), main_proto (cls,128), bg_proto (1,128), y (B,N) int64) -> (pred (B,N) int64, loss) of the
    reference's training forward (capl.py:194-242 with CrossEntropyLoss(ignore_index) as the criterion), one fused HIP
    pipeline forward and backward.  Under block-sharded data parallelism the fake-novel feature sums (forward) and their
    gradients (backward) are all-reduced between the two phases of each pass, as the differentiable all-reduce of the
    unfused form did."""

    @staticmethod
    def forward(ctx, feat, main_proto, bg_proto, y, fake, base_num, ignore_index):
        _need_gpu(feat, "point_feat")
        _need_gpu(main_proto, "main_proto")
        _need_gpu(bg_proto, "bg_proto")
        if y.dtype != torch.int64 or not y.is_cuda:
            raise RuntimeError("gfs3d.capl_train_head: labels must be an int64 HIP tensor")
        feat, main_proto, bg_proto, y = feat.contiguous(), main_proto.contiguous(), bg_proto.contiguous(), y.contiguous()
        B, C, N = feat.shape
        classes = main_proto.shape[0]
        fake = [int(c) for c in fake]
        if not capl_head_supported(feat, classes, len(fake)):
            raise NotImplementedError("gfs3d.capl_train_head: unsupported shape (main_dim 128, <= 23 classes)")
        dev = feat.device
        a = _lib.CaplHeadArgs()
        a.B, a.C, a.N = B, C, N
        a.feat, a.feat_bstride = feat.data_ptr(), C * N
        a.label, a.ignore_index = y.data_ptr(), int(ignore_index)
        a.main_proto, a.classes = main_proto.data_ptr(), classes
        a.bg_proto, a.base_num = bg_proto.data_ptr(), int(base_num)
        a.n_fake = len(fake)
        for i, c in enumerate(fake):
            a.fake[i] = c
        a.half = B // 2
        fsum = torch.zeros((HEAD_MAX_FAKE, C + 1), dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        pred = torch.empty((B, N), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            L = _lib.lib()
            ws = torch.empty(L.gfs3d_capl_head_workspace_bytes(B, N, classes), dtype=torch.uint8, device=dev)   # lives until backward
            a.fsum, a.workspace, a.workspace_bytes = fsum.data_ptr(), ws.data_ptr(), ws.numel()
            a.loss, a.pred = loss.data_ptr(), pred.data_ptr()
            a.stream = torch.cuda.current_stream().cuda_stream
            _lib.check(L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 0), "gfs3d_capl_head_forward_phase(0)")
            if fake and _dp_world() > 1:
                torch.distributed.all_reduce(fsum)
            _lib.check(L.gfs3d_capl_head_forward_phase(ctypes.byref(a), 1), "gfs3d_capl_head_forward_phase(1)")
        # tensor inputs through save_for_backward (autograd's version counters catch an in-place update of the features or
        # prototypes between forward and backward); the scalar fields of the args struct are kept, its pointers are rebuilt
        ctx.save_for_backward(feat, main_proto, bg_proto, y, fsum, loss, ws)
        ctx.scalars = (B, C, N, int(ignore_index), classes, int(base_num), fake)
        ctx.args_blob = bytes(a)
        ctx.has_fake = bool(fake)
        ctx.mark_non_differentiable(pred)
        return pred, loss[0]

    @staticmethod
    def backward(ctx, _dpred, dloss):
        feat, main_proto, bg_proto, y, fsum, loss, ws = ctx.saved_tensors
        B, C, N, ignore_index, classes, base_num, fake = ctx.scalars
        dev = feat.device
        dloss = dloss.to(device=dev, dtype=torch.float32).contiguous()
        dfeat = torch.empty_like(feat)
        dmain = torch.empty_like(main_proto)
        dbg = torch.empty_like(bg_proto)
        dfsum = torch.zeros((HEAD_MAX_FAKE, C + 1), dtype=torch.float32, device=dev)
        # a struct of its own (a byte copy of the forward's scalar fields): backward is re-entrant and never sees a stale pointer
        a = _lib.CaplHeadArgs.from_buffer_copy(ctx.args_blob)
        a.feat, a.label, a.main_proto, a.bg_proto = feat.data_ptr(), y.data_ptr(), main_proto.data_ptr(), bg_proto.data_ptr()
        a.fsum, a.workspace, a.workspace_bytes = fsum.data_ptr(), ws.data_ptr(), ws.numel()
        a.loss, a.pred = loss.data_ptr(), None     # the predictions belong to the caller and are not read by the backward
        a.dloss, a.dfsum = dloss.data_ptr(), dfsum.data_ptr()
        a.dfeat, a.dmain, a.dbg = dfeat.data_ptr(), dmain.data_ptr(), dbg.data_ptr()
        with torch.cuda.device(dev):
            L = _lib.lib()
            a.stream = torch.cuda.current_stream().cuda_stream
            _lib.check(L.gfs3d_capl_head_backward_phase(ctypes.byref(a), 0), "gfs3d_capl_head_backward_phase(0)")
            if ctx.has_fake and _dp_world() > 1:
                torch.distributed.all_reduce(dfsum)
            _lib.check(L.gfs3d_capl_head_backward_phase(ctypes.byref(a), 1), "gfs3d_capl_head_backward_phase(1)")
        return dfeat, dmain, dbg, None, None, None, None


def capl_eval_head(feat, main_proto, gened_proto, word, coding, th, base_num, y=None):
    """capl.forward(eval_model=True) after getFeatures (capl.py:170-192) as one fused HIP pipeline: (point_feat (B,128,N),
    main_proto, gened_proto (cls,128), word (B,N) int32, coding (cls,Kw), th) -> (logits * geometric-word weight (B,cls,N),
    gp_acc, gp_novel_acc).  y (B,N) int64 or None (both accuracies 0., as in get_gp_weight)."""
    _need_gpu(feat, "point_feat")
    _need_gpu(main_proto, "main_proto")
    _need_gpu(gened_proto, "gened_proto")
    _need_gpu(coding, "coding")
    if word.dtype != torch.int32 or not word.is_cuda:
        raise RuntimeError("gfs3d.capl_eval_head: the word index must be an int32 HIP tensor")
    feat, main_proto, gened_proto, coding, word = (t.contiguous() for t in (feat, main_proto, gened_proto, coding, word))
    B, C, N = feat.shape
    classes, Kw = coding.shape
    if not (capl_head_supported(feat, classes, 0) and tuple(main_proto.shape) == (classes, C) and tuple(gened_proto.shape) == (classes, C)
            and classes * Kw <= 16384 and tuple(word.shape) == (B, N)):
        raise NotImplementedError("gfs3d.capl_eval_head: unsupported shape")
    dev = feat.device
    a = _lib.CaplEvalArgs()
    a.B, a.C, a.N = B, C, N
    a.feat, a.feat_bstride = feat.data_ptr(), C * N
    a.main_proto, a.classes, a.base_num = main_proto.data_ptr(), classes, int(base_num)
    a.gened_proto, a.word, a.coding, a.Kw, a.th = gened_proto.data_ptr(), word.data_ptr(), coding.data_ptr(), Kw, float(th)
    if y is not None:
        if y.dtype != torch.int64 or not y.is_cuda:
            raise RuntimeError("gfs3d.capl_eval_head: labels must be an int64 HIP tensor")
        y = y.contiguous()
        a.label = y.data_ptr()
    logits = torch.empty((B, classes, N), dtype=torch.float32, device=dev)
    stats = torch.empty(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L = _lib.lib()
        ws = _workspace(dev, L.gfs3d_capl_head_workspace_bytes(B, N, classes))
        a.logits, a.stats, a.workspace, a.workspace_bytes = logits.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel()
        a.stream = torch.cuda.current_stream().cuda_stream
        _lib.check(L.gfs3d_capl_head_eval(ctypes.byref(a)), "gfs3d_capl_head_eval")
    if y is None:
        return logits, 0., 0.
    return logits, stats[0], stats[1]


def label_presence_begin(y: torch.Tensor):
    """Queue the label-presence kernel and the 8-byte copy to pinned host memory; label_presence_end() waits for that copy only.
    Between the two the caller can queue work that does not depend on the answer (capl.forward queues the whole backbone), so
    the host's wait hides behind device work instead of leaving the device idle while the host reacts."""
    if not (y.is_cuda and y.dtype == torch.int64):
        raise RuntimeError("gfs3d.label_presence: labels must be an int64 HIP tensor")
    y = y.contiguous()
    mask = torch.empty(1, dtype=torch.int64, device=y.device)
    with torch.cuda.device(y.device):
        _lib.check(_lib.lib().gfs3d_label_presence(_p(y), y.numel(), _p(mask), _stream()), "gfs3d_label_presence")
        host = torch.empty(1, dtype=torch.int64, pin_memory=True)
        host.copy_(mask, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
    return host, done, mask            # (mask: kept alive until the copy has run)


def label_presence_end(handle):
    host, done, _ = handle
    done.synchronize()
    m = int(host.item()) & 0xFFFFFFFFFFFFFFFF
    return [c for c in range(64) if (m >> c) & 1]


def label_presence(y: torch.Tensor):
    """Sorted list of the distinct labels in [0, 64) of an int64 HIP tensor: one small kernel (integer OR) and an 8-byte
    read-back instead of torch.unique's device sort (generate_fake_proto's `unique(y)`, capl.py:378-386)."""
    return label_presence_end(label_presence_begin(y))


def capl_train_head(feat, main_proto, bg_proto, y, fake, base_num, ignore_index=-100):
    return CaplTrainHeadFunction.apply(feat, main_proto, bg_proto, y, fake, base_num, ignore_index)


def kmeans_assign(pts: torch.Tensor, cent: torch.Tensor, fast: bool = False, want_sums: bool = True):
    """One Lloyd assignment step on device: labels, per-centroid sums/counts and
    inertia (get_basis.py:210 runs sklearn KMeans on the CPU).  fast=True: the screened form of the Lloyd loop
    (gfs3d_kmeans_assign_fast_f32: same labels / sums / counts, inertia from the screened scores).  want_sums=False (fast form
    only): the labels alone, (label, None, None, None) - the caller moves its running sums with `KMeansDelta`."""
    _need_gpu(pts, "pts")
    _need_gpu(cent, "cent")
    pts, cent = pts.contiguous(), cent.contiguous()
    M, C = pts.shape
    K = cent.shape[0]
    dev = pts.device
    label = torch.empty(M, dtype=torch.int32, device=dev)
    sums = torch.zeros((K, C), dtype=torch.float32, device=dev) if want_sums else None
    # want_sums=False: the kernel still adds into counts / inertia, nobody reads them - no zero fills (two launches per Lloyd iteration)
    counts = torch.zeros(K, dtype=torch.float32, device=dev) if want_sums else torch.empty(K, dtype=torch.float32, device=dev)
    inertia = torch.zeros(1, dtype=torch.float64, device=dev) if want_sums else torch.empty(1, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        L = _lib.lib()
        nbytes = L.gfs3d_kmeans_workspace_bytes(M, C, K)       # > 0: centroid sums as sorted segment sums (fixed order, no atomics)
        ws = _workspace(dev, nbytes) if nbytes else None
        fn = L.gfs3d_kmeans_assign_fast_f32 if (fast and ws is not None) else L.gfs3d_kmeans_assign_ws_f32
        if sums is None and fn is not L.gfs3d_kmeans_assign_fast_f32:
            raise RuntimeError("gfs3d.kmeans_assign: want_sums=False needs the screened form (fast=True, a supported shape)")
        rc = fn(_p(pts), M, C, _p(cent), K, _p(label), _p(sums), _p(counts), _p(inertia), _p(ws), ws.numel() if ws is not None else 0, _stream())
    _lib.check(rc, "gfs3d_kmeans_assign_ws_f32")
    if not want_sums:
        return label, None, None, None
    return label, sums, counts, inertia


def kmeanspp_seed(pts: torch.Tensor, K: int, first: int, u, prune: bool = True):
    """k-means++ seeding on the device (gfs3d_kmeanspp_seed_f32; the `init='k-means++'` of get_basis.py:210): pts (M,C)
    fp32, `first` the first centre's index, `u` the K - 1 uniform variates of the picks (host array: numpy RandomState draws).
    Returns (centers (K,C), picks (K,) int64) - device tensors, nothing waits for the host."""
    _need_gpu(pts, "pts")
    pts = pts.contiguous()
    M, C = pts.shape
    dev = pts.device
    u = torch.as_tensor(u, dtype=torch.float64).reshape(-1)
    if u.numel() != K - 1:
        raise ValueError("gfs3d.kmeanspp_seed: K - 1 uniform variates are needed")
    u_dev = (u if u.numel() else torch.zeros(1, dtype=torch.float64)).to(dev)
    centers = torch.empty((K, C), dtype=torch.float32, device=dev)
    picks = torch.empty(K, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        L = _lib.lib()
        nbytes = L.gfs3d_kmeanspp_workspace_bytes(M, K)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        rc = L.gfs3d_kmeanspp_seed_f32(_p(pts), M, C, K, int(first), _p(u_dev), _p(centers), _p(picks), _p(ws), ws.numel(),
                                       1 if prune else 0, _stream())
    _lib.check(rc, "gfs3d_kmeanspp_seed_f32")
    return centers, picks


KPP_MAX_TRIALS = 8        # KPP_TMAX of csrc/kmeanspp.hip


def kmeanspp_greedy_supported(pts: torch.Tensor, n_trials: int) -> bool:
    """The shapes gfs3d_kmeanspp_greedy_f32 takes: fp32 rows of C = 4 m <= 256 channels, 1..8 candidates per pick
    (sklearn's 2 + int(log K) reaches 9 at K = 1097)."""
    C = int(pts.shape[1])
    return pts.is_cuda and pts.dtype == torch.float32 and C >= 4 and C % 4 == 0 and C <= 256 and 1 <= int(n_trials) <= KPP_MAX_TRIALS


def kmeanspp_greedy(pts: torch.Tensor, K: int, first: int, u, n_trials: int, prune: bool = True):
    """sklearn's greedy k-means++ on the device (gfs3d_kmeanspp_greedy_f32; what `KMeans(init='k-means++')` of get_basis.py:210
    runs): pts (M,C) fp32, `first` the first centre's index, `u` (K - 1, n_trials) uniform variates in sklearn's consumption
    order (host array).  Returns (centers (K,C), picks (K,) int64) - device tensors, nothing waits for the host."""
    _need_gpu(pts, "pts")
    pts = pts.contiguous()
    M, C = pts.shape
    dev = pts.device
    u = torch.as_tensor(u, dtype=torch.float64).reshape(-1)
    if u.numel() != (K - 1) * n_trials:
        raise ValueError("gfs3d.kmeanspp_greedy: (K - 1) * n_trials uniform variates are needed")
    u_dev = (u if u.numel() else torch.zeros(1, dtype=torch.float64)).to(dev)
    centers = torch.empty((K, C), dtype=torch.float32, device=dev)
    picks = torch.empty(K, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        L = _lib.lib()
        nbytes = L.gfs3d_kmeanspp_greedy_workspace_bytes(M, C, K)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        rc = L.gfs3d_kmeanspp_greedy_f32(_p(pts), M, C, K, int(first), _p(u_dev), int(n_trials), _p(centers), _p(picks), _p(ws), ws.numel(),
                                         1 if prune else 0, _stream())
    _lib.check(rc, "gfs3d_kmeanspp_greedy_f32")
    return centers, picks


class KMeansDelta:
    """Running centroid sums of a Lloyd loop that FOLLOW the labels (gfs3d_kmeans_delta_fix): after a full pass `load(sums,
    counts)`, then per iteration `changed = move(label_old, label_new)` touches only the points whose label differs, and
    `sums_counts()` returns the fp32 (sums, counts).  The sums live as 64-bit fixed-point integers whose scale is fixed at
    construction from max |x| and M (no sum can overflow); integer atomics make them independent of the order of arrival."""

    def __init__(self, pts: torch.Tensor, K: int):
        _need_gpu(pts, "pts")
        self.pts = pts.contiguous()
        self.M, self.C = self.pts.shape
        self.K = int(K)
        lo, hi = torch.aminmax(self.pts)                               # no full-size |x| temporary (1.6 GB at configs[2] size)
        peak = float(torch.maximum(-lo, hi))                           # one read at construction
        import math
        bits_x = max(1, math.ceil(math.log2(max(peak, 1e-30))) + 1)
        self.shift = int(max(0, min(40, 62 - math.ceil(math.log2(max(self.M, 2))) - bits_x)))
        dev = self.pts.device
        self.sums_fix = torch.zeros((self.K, self.C), dtype=torch.int64, device=dev)
        self.counts_fix = torch.zeros(self.K, dtype=torch.int64, device=dev)
        self.changed = torch.zeros(1, dtype=torch.int64, device=dev)
        self._changed_is_zero = True                                   # (step() resets the counter on the device)

    def load(self, sums: torch.Tensor, counts: torch.Tensor):
        with torch.cuda.device(self.pts.device):
            rc = _lib.lib().gfs3d_kmeans_fix_from_f32(_p(sums.contiguous()), _p(counts.contiguous()), self.K, self.C, self.shift,
                                                      _p(self.sums_fix), _p(self.counts_fix), _stream())
        _lib.check(rc, "gfs3d_kmeans_fix_from_f32")

    def move(self, label_old: torch.Tensor, label_new: torch.Tensor) -> torch.Tensor:
        """-> device int64[1]: the number of points whose label changed (their rows were moved between the sums)."""
        if label_old.dtype != torch.int32 or label_new.dtype != torch.int32:
            raise RuntimeError("gfs3d.KMeansDelta: labels must be int32 device tensors")
        if not self._changed_is_zero:
            self.changed.zero_()
        self._changed_is_zero = False
        with torch.cuda.device(self.pts.device):
            rc = _lib.lib().gfs3d_kmeans_delta_fix(_p(self.pts), self.M, self.C, self.K, _p(label_old), _p(label_new), self.shift,
                                                   _p(self.sums_fix), _p(self.counts_fix), _p(self.changed), _stream())
        _lib.check(rc, "gfs3d_kmeans_delta_fix")
        return self.changed

    def step(self, cent_old: torch.Tensor):
        """The tail of a Lloyd iteration in one launch -> (sums, counts, cent_new, state): the fp32 sums / counts, the new centres
        (member means, an empty cluster keeps its old centre) and state = float32[3] {squared centre shift, labels changed by the
        last `move`, empty clusters} on the device (gfs3d_kmeans_fix_step); the changed-label counter is reset by the kernel."""
        dev = self.pts.device
        cent_old = cent_old.contiguous()
        sums = torch.empty((self.K, self.C), dtype=torch.float32, device=dev)
        counts = torch.empty(self.K, dtype=torch.float32, device=dev)
        new = torch.empty((self.K, self.C), dtype=torch.float32, device=dev)
        state = torch.empty(3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().gfs3d_kmeans_fix_step(_p(self.sums_fix), _p(self.counts_fix), self.K, self.C, self.shift, _p(cent_old),
                                                  _p(self.changed), _p(sums), _p(counts), _p(new), _p(state), _stream())
        _lib.check(rc, "gfs3d_kmeans_fix_step")
        self._changed_is_zero = True
        return sums, counts, new, state

    def sums_counts(self):
        dev = self.pts.device
        sums = torch.empty((self.K, self.C), dtype=torch.float32, device=dev)
        counts = torch.empty(self.K, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().gfs3d_kmeans_fix_to_f32(_p(self.sums_fix), _p(self.counts_fix), self.K, self.C, self.shift, _p(sums), _p(counts),
                                                    _stream())
        _lib.check(rc, "gfs3d_kmeans_fix_to_f32")
        return sums, counts

"""Block-sharded data parallelism helpers (one process per GPU, RCCL over xGMI
via torch.distributed backend 'nccl'; 'gloo' on CPU for the logic tests).

The reference is single-device (no DP/DDP anywhere), so this
is new functionality defined by BASELINE.json: blocks are independent units, so
ranks own disjoint shards of the block batch and the only data-path exchange per
training step is ONE flat-bucket all-reduce of the gradients (404,544 fp32 =
1.6 MB for the CAPL model, 261,504 for the bare backbone): at that size a ring is
latency-bound, so a single bucket (one collective per step) is the right shape
for point-to-point xGMI links.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> Tuple[int, int]:
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_range(n_items: int, rank: Optional[int] = None, world_size: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for this rank; shard sizes differ by at most 1."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x, differentiable: dx = sum over ranks of dy (every rank's y depends on every rank's x)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.contiguous().clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, dy):
        g = dy.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g, None


def all_reduce_sum_diff(x: torch.Tensor, group=None) -> torch.Tensor:
    """Differentiable SUM all-reduce (generate_fake_proto's (feature sum, count) under block-sharded DP).  The same
    arithmetic as torch.distributed.nn.functional.all_reduce - which is NOT used: with torch 2.10 a process that has imported
    torch.distributed.nn (the import alone, no call) aborts at interpreter exit every other run ("terminate called without
    an active exception": bisected in the 8-rank gloo rehearsal of bench.py, round 5)."""
    return _AllReduceSum.apply(x, group)


class FlatGradBucket:
    """Data-parallel gradient exchange over ONE flat fp32 buffer: after backward the parameter gradients are
    packed by a single multi-tensor copy, summed over the ranks, and handed back to the parameters as views of
    the buffer (no per-parameter collectives, no unpack copies).  `zero()` drops the previous gradients so
    autograd writes each one once instead of launching an accumulate kernel per parameter.

    `early`: parameters whose gradients are complete long before backward ends (the point-wise stack of DGCNN:
    88 % of the parameters, first in backward order).  Their slice of the buffer is all-reduced asynchronously
    from a post-accumulate hook as soon as the last of them has its gradient, i.e. underneath the EdgeConv
    backward; `all_reduce_mean()` then only exchanges the small remainder and waits for the early slice.

    `overlap=False` keeps the early slice in the flat layout but launches it from `all_reduce_mean()`: REQUIRED when
    other collectives run inside backward on the same communicator (synchronised BatchNorm: gfs3d.ops.BNSync /
    torch SyncBatchNorm) - a hook that fires on one rank and not on another would otherwise put the slice before the
    BatchNorm all-reduces on the first and after them on the second."""

    def __init__(self, params: Iterable[torch.nn.Parameter], early: Optional[Iterable[torch.nn.Parameter]] = None,
                 overlap: bool = True):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("FlatGradBucket: no trainable parameters")
        early_ids = {id(p) for p in (early or []) if p.requires_grad}
        self.early: List[torch.nn.Parameter] = [p for p in params if id(p) in early_ids]
        self.params: List[torch.nn.Parameter] = self.early + [p for p in params if id(p) not in early_ids]
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=dt, device=dev)
        self.views: List[torch.Tensor] = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.n_early = sum(p.numel() for p in self.early)
        self._seen = 0
        self._work = None
        self.early_launches = 0                      # how often the overlapped path ran (tests)
        self.force_collectives = False               # tests: issue the collectives even in a 1-rank group
        self.overlap = bool(overlap)
        if self.early and self.overlap and is_dist():
            for p in self.early:
                p.register_post_accumulate_grad_hook(self._on_early_grad)

    def zero(self) -> None:
        for p in self.params:
            p.grad = None
        self._seen = 0
        self._work = None

    def _pack(self, lo: int, hi: int) -> None:
        views, params = self.views[lo:hi], self.params[lo:hi]
        have = [(v, p.grad) for v, p in zip(views, params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        for v, p in zip(views, params):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g_ for _, g_ in have])

    def _on_early_grad(self, _param) -> None:
        if dist.get_world_size() == 1 and not self.force_collectives:
            return
        self._seen += 1
        if self._seen == len(self.early) and self._work is None:
            self._pack(0, len(self.early))
            self._work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)
            self.early_launches += 1

    def all_reduce_mean(self) -> None:
        """SUM over ranks then divide by world size (mean over the global batch when every
        rank's loss is a mean over its equally sized shard)."""
        if not is_dist() or (dist.get_world_size() == 1 and not self.force_collectives):
            return
        if self.early:
            # Every rank issues the SAME two collectives in the same order (early slice, remainder) whether or not its
            # hook fired: a rank on which an early parameter received no gradient launches the slice late, here.
            if self._work is None:
                self._pack(0, len(self.early))
                self._work = dist.all_reduce(self.flat[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)
            self._pack(len(self.early), len(self.params))
            if self.n_early < self.flat.numel():
                dist.all_reduce(self.flat[self.n_early:], op=dist.ReduceOp.SUM)
            self._work.wait()
            self._work = None
            self._seen = 0                       # the next backward counts from zero even if zero() is skipped
        else:
            self._pack(0, len(self.params))
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        for v, p in zip(self.views, self.params):
            p.grad = v


def all_reduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """Integer counters (confusion counts for mIoU, word histograms): exact SUM over ranks."""
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def enable_sync_batchnorm(model: torch.nn.Module, group=None) -> torch.nn.Module:
    """Whole-batch BatchNorm statistics under block-sharded DP (what the single-device reference
    computes): the EdgeConv kernels exchange their fp64 sums between phases (gfs3d.ops.BNSync), the
    BatchNorm1d layers of the point-wise heads become torch SyncBatchNorm.  Returns the model."""
    from . import ops
    from model.dgcnn import DGCNN
    sync = ops.BNSync.from_group(group)
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model, process_group=group)
    for m in model.modules():
        if isinstance(m, DGCNN):
            m.bn_sync = sync
    return model

"""Morton (Z-curve) order of the points of a block: consecutive points are spatial neighbours, so the rows a tile of queries
gathers (its neighbours' projections, the candidates of its kNN, the targets of its edges) are few and close together.
Measured on the bench's synthetic blocks (scripts/exp_morton_locality.py): 59 % of a point's 20 neighbours lie within 64
positions of it (11 % in the sampled order), a 64-query tile touches 327 distinct rows instead of 940.

MEASURED NEUTRAL (round 5, DESIGN_HISTORY.md): delivering the c2 blocks in this order leaves the gather kernels where they were
(their rows are L2-resident per XCD either way: ec_stats1 44.5 vs 43.4 us, ec_main_fwd 104.1 vs 103.5, ec_main_bwd_pair 217.6 vs
213.6, csr_gather 91.4 vs 90.8). The first run of the experiment also broke the screened kNN (its buckets were runs of consecutive
points, so the neighbours of a sorted cloud shared buckets and the lists overflowed: knn_exact_kernel 45 -> 775 us); csrc/knn.hip
now strides and rotates its stages over the cloud and is indifferent to the order (42 vs 44 us). Kept as the reproduction knob of
the experiment (`GFS3D_BENCH_MORTON=1 python bench.py`, scripts/exp_morton_locality.py); the product path keeps the sampled order
the reference's loader produces (dataloaders/loader.py:64-82)."""
from __future__ import annotations

import torch


def morton_order(xyz: torch.Tensor, bits: int = 10):
    """xyz (B,3,N) -> (order (B,N), rank (B,N)): `order[b]` lists the points of block b along the Morton curve of its bounding
    box (ties: ascending index), `rank` is the inverse permutation."""
    lo = xyz.amin(dim=2, keepdim=True)
    span = (xyz.amax(dim=2, keepdim=True) - lo).clamp_min(1e-12)
    q = ((xyz - lo) / span * float((1 << bits) - 1)).long().clamp_(0, (1 << bits) - 1)
    code = torch.zeros(xyz.shape[0], xyz.shape[2], dtype=torch.long, device=xyz.device)
    for bit in range(bits):
        for ax in range(3):
            code |= ((q[:, ax] >> bit) & 1) << (3 * bit + ax)
    order = torch.argsort(code, dim=1, stable=True)
    rank = torch.empty_like(order)
    rank.scatter_(1, order, torch.arange(xyz.shape[2], device=xyz.device).expand_as(order))
    return order, rank

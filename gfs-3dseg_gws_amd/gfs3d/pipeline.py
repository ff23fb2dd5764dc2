"""Block sampling and normalisation on the device.

The reference builds every training block on the host: np.load of an (M, 7|8) float array
[x y z r g b label (segment)], a numpy draw of 2048 point indices, min-shift / optional augmentation / unit-box
normalisation, label remapping - one Python call per block in 16 loader workers (dataloaders/loader.py:39-151).
At the block rates of the HIP path that loader is the bottleneck, so here the blocks live in HBM and a whole batch
is sampled, gathered and normalised by a handful of device-wide torch ops:

    sampler = BlockSampler(blocks)                    # list of (M_i, 7|8) tensors, moved to the device once
    x, y, seg = sampler.batch(block_ids, num_point=2048, sampled_classes=[...])      # (B,9,N), (B,N), (B,N)

Same arithmetic as the reference given the same point indices (`assemble`), same sampling distributions
(`sample_indices`; the draws come from a torch generator, not numpy's stream).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch


def sample_indices(labels: torch.Tensor, num_point: int, sampled_class: Optional[int] = None, random_sample: bool = False,
                   generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Point indices of one block (loader.py:67-82) on labels' device."""
    n = labels.shape[0]
    dev = labels.device

    def draw(pool_size: int, count: int, replace: bool) -> torch.Tensor:
        if count <= 0:
            return torch.empty(0, dtype=torch.long, device=dev)
        if replace:
            return torch.randint(0, pool_size, (count,), device=dev, generator=generator)
        return torch.randperm(pool_size, device=dev, generator=generator)[:count]

    if random_sample:
        return draw(n, num_point, n < num_point)
    valid = torch.nonzero(labels == sampled_class, as_tuple=False).flatten()
    n_valid = int(valid.numel()) if n < num_point else int(valid.numel() / float(n) * num_point)
    a = valid[draw(int(valid.numel()), n_valid, False)]
    b = draw(n, num_point - n_valid, n < num_point)
    return torch.cat([a, b])


def augment_draws(cfg: dict, num_point: int, device, generator: Optional[torch.Generator] = None):
    """The random quantities of augment_pointcloud (loader.py:132-151): a 3x3 matrix (zoom, rotation about z,
    x / y mirrors) and, if cfg['jitter'], clip(0.01 randn, +-0.05) per coordinate."""
    def u():
        return float(torch.rand((), generator=generator, device="cpu" if generator is None or generator.device.type == "cpu" else device))
    m = torch.eye(3, dtype=torch.float64)
    if cfg.get("scale", 0) > 1:
        s = 1.0 / cfg["scale"] + u() * (cfg["scale"] - 1.0 / cfg["scale"])
        m = m * s
    if cfg.get("rot", 0) == 1:
        ang = u() * 2.0 * math.pi
        c, s_ = math.cos(ang), math.sin(ang)
        m = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64) @ m
    if cfg.get("mirror_prob", 0) > 0:
        if u() < cfg["mirror_prob"] / 2:
            m = torch.diag(torch.tensor([-1.0, 1.0, 1.0], dtype=torch.float64)) @ m
        if u() < cfg["mirror_prob"] / 2:
            m = torch.diag(torch.tensor([1.0, -1.0, 1.0], dtype=torch.float64)) @ m
    jit = None
    if cfg.get("jitter", 0):
        jit = (0.01 * torch.randn((num_point, 3), device=device, generator=generator if generator is not None and generator.device == torch.device(device) else None)).clamp(-0.05, 0.05).float()
    return m.to(device), jit


def assemble(rows: torch.Tensor, pc_attribs: str = "xyzrgbXYZ", sampled_classes: Sequence[int] = (), sampled_class: int = 0,
             support: bool = False, use_all_classes: bool = False, augm_matrix: Optional[torch.Tensor] = None,
             jitter: Optional[torch.Tensor] = None):
    """Gathered rows (..., n, 7|8) -> (ptcloud (..., 3A, n) fp32 channel-first as the Dataset classes return it,
    groundtruth (..., n) int64, segment_label (..., n) fp32)   (loader.py:84-129, 200)."""
    xyz = rows[..., 0:3]
    xyz = xyz - xyz.amin(dim=-2, keepdim=True)
    if augm_matrix is not None:
        xyz = xyz @ augm_matrix.to(xyz.dtype).transpose(-1, -2)
    if jitter is not None:
        xyz = xyz + jitter
    parts = []
    if "xyz" in pc_attribs:
        parts.append(xyz)
    if "rgb" in pc_attribs:
        # (a tensor divisor: torch turns division by a Python scalar into a multiplication by its reciprocal)
        parts.append(rows[..., 3:6] / torch.full((), 255.0, dtype=rows.dtype, device=rows.device))
    if "XYZ" in pc_attribs:
        XYZ = xyz - xyz.amin(dim=-2, keepdim=True)
        parts.append(XYZ / XYZ.amax(dim=-2, keepdim=True))
    pc = torch.cat(parts, dim=-1).transpose(-1, -2).float().contiguous()
    labels = rows[..., 6].long()
    if support:
        gt = (labels == sampled_class).long()
    else:
        # table lookup instead of the per-point Python loop (loader.py:108-112): class id -> position (+1: 0 = background)
        off = 0 if use_all_classes else 1
        hi = int(max([int(labels.max())] + [int(c) for c in sampled_classes])) + 1
        table = torch.zeros(hi, dtype=torch.long, device=rows.device)
        for i, c in enumerate(sampled_classes):
            table[int(c)] = i + off
        gt = table[labels]
    seg = rows[..., 7].float() if rows.shape[-1] == 8 else torch.zeros(rows.shape[:-1], dtype=torch.float32, device=rows.device)
    return pc, gt, seg


class BlockSampler:
    """Blocks resident in HBM; `batch()` = sample_pointcloud for a list of block ids, one gather + one assemble."""

    def __init__(self, blocks: List[torch.Tensor], device="cuda"):
        self.blocks = [b.to(device) for b in blocks]
        self.device = torch.device(device)

    def batch(self, block_ids: Sequence[int], num_point: int = 2048, pc_attribs: str = "xyzrgbXYZ",
              sampled_classes: Sequence[int] = (), sampled_class: Optional[int] = None, support: bool = False,
              use_all_classes: bool = False, pc_augm: bool = False, pc_augm_config: Optional[dict] = None,
              generator: Optional[torch.Generator] = None):
        rows = []
        for bid in block_ids:
            blk = self.blocks[int(bid)]
            idx = sample_indices(blk[:, 6], num_point, sampled_class, random_sample=sampled_class is None, generator=generator)
            rows.append(blk[idx])
        cols = {r.shape[1] for r in rows}
        if len(cols) != 1:
            raise RuntimeError("gfs3d.pipeline: blocks of one batch must have the same number of columns (7 or 8)")
        rows = torch.stack(rows, dim=0)                                           # (B, n, 7|8)
        m = jit = None
        if pc_augm:
            ms, js = zip(*[augment_draws(pc_augm_config, num_point, self.device, generator) for _ in block_ids])
            m = torch.stack(ms, dim=0)
            jit = torch.stack(js, dim=0) if js[0] is not None else None
        return assemble(rows, pc_attribs, sampled_classes, 0 if sampled_class is None else sampled_class, support,
                        use_all_classes, m, jit)

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


def _usable_generator(generator: Optional[torch.Generator], device) -> Optional[torch.Generator]:
    """`generator` if it can drive draws on `device`, None if none was given.  torch.device('cuda') and a CUDA generator's
    'cuda:0' name the same card when 0 is the current device: the index is normalised before the comparison.  A generator on
    another device (a CPU generator for device draws) RAISES - silently drawing from the global stream instead made seeded runs
    irreproducible without an error."""
    if generator is None:
        return None
    dev, gdev = torch.device(device), generator.device
    if dev.type == gdev.type:
        if dev.type != "cuda":
            return generator
        cur = torch.cuda.current_device()
        if (dev.index if dev.index is not None else cur) == (gdev.index if gdev.index is not None else cur):
            return generator
    raise RuntimeError(f"gfs3d.pipeline: the generator lives on {gdev} but the draws happen on {dev}; pass torch.Generator(device=...) "
                       "of the sampler's device")


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
        jit = (0.01 * torch.randn((num_point, 3), device=device,
                                  generator=generator if generator is not None and generator.device.type != "cpu" else None)).clamp(-0.05, 0.05).float()
    return m.to(device), jit


def augment_draws_batch(cfg: dict, B: int, num_point: int, device, generator: Optional[torch.Generator] = None):
    """augment_draws for a whole batch in tensor ops (no host reads): (B,3,3) float64 matrices
    M = mirror_y mirror_x R_z(angle) zoom(s) (the reference's order of composition, loader.py:134-145) and the (B,n,3) jitter."""
    gen = _usable_generator(generator, device)
    u = torch.rand((B, 4), dtype=torch.float64, device=device, generator=gen)
    one, zero = torch.ones(B, dtype=torch.float64, device=device), torch.zeros(B, dtype=torch.float64, device=device)
    s = one
    if cfg.get("scale", 0) > 1:
        s = 1.0 / cfg["scale"] + u[:, 0] * (cfg["scale"] - 1.0 / cfg["scale"])
    c, s_ = one, zero
    if cfg.get("rot", 0) == 1:
        ang = u[:, 1] * (2.0 * math.pi)
        c, s_ = torch.cos(ang), torch.sin(ang)
    sx = sy = one
    if cfg.get("mirror_prob", 0) > 0:
        sx = torch.where(u[:, 2] < cfg["mirror_prob"] / 2, -one, one)
        sy = torch.where(u[:, 3] < cfg["mirror_prob"] / 2, -one, one)
    m = torch.stack([torch.stack([sx * c, -sx * s_, zero], 1), torch.stack([sy * s_, sy * c, zero], 1),
                     torch.stack([zero, zero, one], 1)], 1) * s[:, None, None]
    jit = None
    if cfg.get("jitter", 0):
        jit = (0.01 * torch.randn((B, num_point, 3), device=device, generator=gen)).clamp(-0.05, 0.05).float()
    return m, jit


def label_lookup_table(sampled_classes: Sequence[int], use_all_classes: bool, num_labels: int, device) -> torch.Tensor:
    """class id -> position in `sampled_classes` (+1 unless use_all_classes: 0 is the background), 0 for every other id
    (loader.py:105-120), as a device tensor of length max(num_labels, largest class id + 1)."""
    off = 0 if use_all_classes else 1
    hi = max([int(num_labels)] + [int(c) + 1 for c in sampled_classes])
    table = torch.zeros(hi, dtype=torch.long)
    for i, c in enumerate(sampled_classes):
        table[int(c)] = i + off
    return table.to(device, non_blocking=True)


def assemble(rows: torch.Tensor, pc_attribs: str = "xyzrgbXYZ", sampled_classes: Sequence[int] = (), sampled_class: int = 0,
             support: bool = False, use_all_classes: bool = False, augm_matrix: Optional[torch.Tensor] = None,
             jitter: Optional[torch.Tensor] = None, num_labels: Optional[int] = None, label_table: Optional[torch.Tensor] = None):
    """Gathered rows (..., n, 7|8) -> (ptcloud (..., 3A, n) fp32 channel-first as the Dataset classes return it,
    groundtruth (..., n) int64, segment_label (..., n) fp32)   (loader.py:84-129, 200).
    num_labels: an upper bound (exclusive) of the label ids in `rows`, known to the caller (BlockSampler reads it once at
    construction); without it the label table is sized from `labels.max()`, a device -> host read.  label_table: the
    class id -> ground-truth id table already on the device (`label_lookup_table`; BlockSampler keeps one per class list)."""
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
        if label_table is None:
            top = int(labels.max()) + 1 if num_labels is None else int(num_labels)
            label_table = label_lookup_table(sampled_classes, use_all_classes, top, rows.device)
        gt = label_table[labels]
    seg = rows[..., 7].float() if rows.shape[-1] == 8 else torch.zeros(rows.shape[:-1], dtype=torch.float32, device=rows.device)
    return pc, gt, seg


class BlockSampler:
    """Blocks resident in HBM as ONE (sum M_i, 7|8) table; `batch()` = sample_pointcloud (loader.py:39-129) for a list of block
    ids as a fixed sequence of batch-wide tensor ops - random keys, two top-k selections, one row gather, one assemble - with
    NO device -> host read (block sizes and the label range are host-side facts taken at construction), so the sampler
    runs ahead of the training step on its stream.  The draws have the reference's distributions: uniform subsets without
    replacement (the num_point smallest of M_i uniform keys, in key order: a uniformly random ordered subset), uniform draws
    with replacement for blocks smaller than num_point, the class share `int(n_class / M_i * num_point)` of loader.py:71-76."""

    def __init__(self, blocks: List[torch.Tensor], device="cuda"):
        self.device = torch.device(device)
        cols = {int(b.shape[1]) for b in blocks}
        if len(cols) != 1:
            raise RuntimeError("gfs3d.pipeline: blocks must have the same number of columns (7 or 8)")
        self.sizes = [int(b.shape[0]) for b in blocks]
        self.offsets = [0]
        for m in self.sizes:
            self.offsets.append(self.offsets[-1] + m)
        self.data = torch.cat([b.to(self.device) for b in blocks], dim=0).contiguous()
        self.blocks = [self.data[o:o + m] for o, m in zip(self.offsets[:-1], self.sizes)]      # views of the table
        self.num_labels = int(self.data[:, 6].max()) + 1                                         # the one read, at construction
        self._tables = {}

    def sample_batch_indices(self, block_ids: Sequence[int], num_point: int, sampled_class: Optional[int] = None,
                             generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """(B, num_point) row numbers into `self.data`: the point draw of sample_pointcloud for every block of the batch."""
        dev = self.device
        gen = _usable_generator(generator, dev)
        B, n = len(block_ids), int(num_point)
        sizes = [self.sizes[int(b)] for b in block_ids]
        mmax = max(sizes)
        kk = min(n, mmax)
        host = torch.tensor([sizes, [self.offsets[int(b)] for b in block_ids]], dtype=torch.long)
        size_t, off_t = host.to(dev, non_blocking=True).unbind(0)
        pos = torch.arange(mmax, device=dev)[None, :]
        inside = pos < size_t[:, None]
        small = (size_t < n)[:, None]                                  # drawn WITH replacement (loader.py:66, 79)
        col = torch.arange(n, device=dev)[None, :]

        def ordered_subset(mask):
            keys = torch.rand((B, mmax), device=dev, generator=gen)
            order = torch.where(mask, keys, torch.full_like(keys, 2.0)).topk(kk, dim=1, largest=False).indices
            if kk < n:                                                 # every block of the batch is small: padded, never read
                order = torch.cat([order, order.new_zeros((B, n - kk))], dim=1)
            return order

        with_repl = (torch.rand((B, n), device=dev, generator=gen).double() * size_t[:, None]).long().clamp_(max=mmax - 1)
        with_repl = torch.minimum(with_repl, size_t[:, None] - 1)
        if sampled_class is None:                                      # random_sample=True
            idx = torch.where(small, with_repl, ordered_subset(inside))
        else:
            flat = (off_t[:, None] + pos).clamp_(max=self.data.shape[0] - 1)
            valid = inside & (self.data[:, 6][flat] == sampled_class)
            nv = valid.sum(dim=1)
            share = (nv.double() / size_t.double() * n).long()         # int(valid_ratio * num_point), the same IEEE operations
            n_valid = torch.where(small[:, 0], nv, share)[:, None]
            first = ordered_subset(valid)
            rest = torch.where(small, with_repl, ordered_subset(inside))
            idx = torch.where(col < n_valid, first, rest.gather(1, (col - n_valid).clamp_(min=0)))
        return off_t[:, None] + idx

    def batch(self, block_ids: Sequence[int], num_point: int = 2048, pc_attribs: str = "xyzrgbXYZ",
              sampled_classes: Sequence[int] = (), sampled_class: Optional[int] = None, support: bool = False,
              use_all_classes: bool = False, pc_augm: bool = False, pc_augm_config: Optional[dict] = None,
              generator: Optional[torch.Generator] = None):
        rows = self.data[self.sample_batch_indices(block_ids, num_point, sampled_class, generator)]      # (B, n, 7|8)
        m = jit = None
        if pc_augm:
            m, jit = augment_draws_batch(pc_augm_config, len(block_ids), num_point, self.device, generator)
        key = (tuple(int(c) for c in sampled_classes), bool(use_all_classes))
        if key not in self._tables:
            self._tables[key] = label_lookup_table(sampled_classes, use_all_classes, self.num_labels, self.device)
        return assemble(rows, pc_attribs, sampled_classes, 0 if sampled_class is None else sampled_class, support,
                        use_all_classes, m, jit, num_labels=self.num_labels, label_table=self._tables[key])

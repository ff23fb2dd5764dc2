"""Synthetic S3DIS-shaped input blocks for benchmarks (SURVEY.md section 8d recipe, mirrors what
dataloaders/loader.py:82-101 produces): xyz in a 1 x 1 x 3 m block, rgb in [0, 1], XYZ = xyz / per-block max."""
import torch


def synthetic_blocks(B: int, N: int = 2048, seed: int = 0) -> torch.Tensor:
    """-> (B, 9, N) fp32 on the CPU (seeded; move it to the device once, outside the timed region)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * torch.tensor([1.0, 1.0, 3.0])
    rgb = torch.rand(B, N, 3, generator=g)
    XYZ = xyz / xyz.amax(dim=1, keepdim=True)
    return torch.cat((xyz, rgb, XYZ), dim=2).transpose(1, 2).contiguous()

"""Test clouds shared by the experiment scripts."""
import torch


def structured(Bq, Nq, seed):
    """A room-like block: points on 6 planes of a box, colour constant per plane + noise (then the loader's 9 attributes)."""
    g = torch.Generator().manual_seed(seed)
    face = torch.randint(0, 6, (Bq, Nq), generator=g)
    uv = torch.rand(Bq, Nq, 2, generator=g)
    xyz = torch.zeros(Bq, Nq, 3)
    for f in range(6):
        ax, side = f // 2, f % 2
        m = face == f
        others = [a for a in range(3) if a != ax]
        xyz[..., ax][m] = float(side)
        xyz[..., others[0]][m] = uv[..., 0][m]
        xyz[..., others[1]][m] = uv[..., 1][m]
    xyz = xyz * torch.tensor([1.0, 1.0, 3.0]) + 0.005 * torch.randn(Bq, Nq, 3, generator=g)
    xyz = xyz - xyz.amin(dim=1, keepdim=True)
    rgb = (torch.rand(6, 3, generator=g)[face] + 0.03 * torch.randn(Bq, Nq, 3, generator=g)).clamp(0, 1)
    XYZ = xyz / xyz.amax(dim=1, keepdim=True)
    return torch.cat([xyz, rgb, XYZ], dim=2).transpose(1, 2).contiguous()

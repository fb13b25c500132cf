"""Seeded synthetic clouds shared by the CPU and GPU tests (no reference data needed)."""
import numpy as np
import torch


def pc_norm(x):
    """centre + unit-sphere normalise, as ShapeNet.pc_norm does (datasets/ShapeNet55Dataset.py:45-51)."""
    x = x - x.mean(dim=1, keepdim=True)
    m = x.pow(2).sum(-1).sqrt().amax(dim=1, keepdim=True).unsqueeze(-1)
    return x / m


def uniform(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return pc_norm(torch.rand(B, N, 3, generator=g) * 2 - 1).contiguous()


def gaussian(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return pc_norm(torch.randn(B, N, 3, generator=g)).contiguous()


def lattice(B, N, seed):
    """integer lattice points scaled by 1/8: exact distance ties everywhere."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randint(-6, 7, (B, N, 3), generator=g).float() / 8).contiguous()


def duplicates(B, N, seed):
    """every point appears ~4 times."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(B, N // 4 + 1, 3, generator=g) * 2 - 1
    idx = torch.randint(0, N // 4 + 1, (B, N), generator=g)
    return torch.gather(base, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()


def near_origin(B, N, seed):
    """a third of the points inside the |p|^2 <= 1e-3 ball that FPS must skip."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    x[:, ::3] *= 0.015
    return x.contiguous()


def all_skipped(B, N, seed):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(B, N, 3, generator=g) * 2 - 1) * 0.01).contiguous()


FAMILIES = {"uniform": uniform, "gaussian": gaussian, "lattice": lattice, "duplicates": duplicates,
            "near_origin": near_origin, "all_skipped": all_skipped}

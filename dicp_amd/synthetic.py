"""Deterministic synthetic scan pairs: the benchmark's input shape.

Per cloud: m target points uniform in [-10,10]^3 with unit normals; the source is n target
rows (random picks) + N(0, 0.01^2) noise, moved by a small fixed SE(3) (rotation <= 0.05 rad
about a seeded axis, translation <= 0.3 m).  Generated on the CPU generator so that the GPU
box and this container produce identical bits."""
import math

import torch


def _rot(axis, ang):
    a = axis / axis.norm()
    K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=torch.float64)
    return torch.eye(3, dtype=torch.float64) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def make_pairs(N, n, m, seed=0, dtype=torch.float32, noise=0.01, max_rot=0.05, max_trans=0.3, device=None, first=0):
    """-> source (N,n,3), target (N,m,6) [xyz, unit normal].  Cloud b is seeded by its GLOBAL index
    first+b, so a sharded job (rank g: first = g*N) draws the same clouds as a single-process one."""
    src = torch.empty((N, n, 3), dtype=torch.float64)
    tgt = torch.empty((N, m, 6), dtype=torch.float64)
    for b in range(N):
        g = torch.Generator().manual_seed(100000 * seed + first + b)
        pts = (torch.rand((m, 3), generator=g, dtype=torch.float64) - 0.5) * 20.0
        nrm = torch.randn((m, 3), generator=g, dtype=torch.float64)
        nrm = nrm / nrm.norm(dim=1, keepdim=True)
        pick = torch.randint(0, m, (n,), generator=g)
        s_t = pts[pick] + noise * torch.randn((n, 3), generator=g, dtype=torch.float64)
        axis = torch.randn(3, generator=g, dtype=torch.float64)
        ang = float(torch.rand(1, generator=g, dtype=torch.float64)) * max_rot
        trans = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 2.0 * max_trans
        C = _rot(axis, ang)
        src[b] = (s_t - trans) @ C            # p = C^T (s - r): ICP must find T_ts = [C, r]
        tgt[b, :, :3] = pts
        tgt[b, :, 3:] = nrm
    src, tgt = src.to(dtype), tgt.to(dtype)
    if device is not None:
        src, tgt = src.to(device), tgt.to(device)
    return src, tgt

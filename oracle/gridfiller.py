"""ORACLE (test infrastructure, never shipped as product): CPU restatement of MeshUDF's
coarse-to-fine UDF grid filling.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Pinned by tests/golden/ (tools/make_golden.py runs the
imported reference GridFiller on an analytic field and on the synthetic decoder).

Written as dense 3-D lattice/block array operations (strided lattice views, a
[n,s,n,s,n,s] block view, 2x repeat of the refine mask) instead of the reference's
materialised flat index tensors; the *result* is defined to be identical:

Reference anchors (relative to /root/reference):
  level list, coordinates        meshudf/meshudf.py:36-78
  per-level query / refine       meshudf/meshudf.py:123-194
  gradient band                  meshudf/meshudf.py:199-206
  dense variant                  meshudf/meshudf.py:254-304
  udf[udf<0]=0                   meshudf/meshudf.py:342
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Tuple

import torch

from .decoder import sample_grads, sample_udf

Tensor = torch.Tensor


def grid_levels(N: int) -> List[int]:
    return [32 * (2 ** i) for i in range(int(math.log2(N) - 4))]


def axis_coords(N: int, origin: float = -1.0, side: float = 2.0) -> Tensor:
    """float32(idx) * float32(side/(N-1)) + origin, two separately rounded fp32 ops."""
    voxel = side / (N - 1)
    return (torch.arange(N, dtype=torch.float32) * voxel) + origin


def refine_threshold(n_level: int) -> float:
    step_size = 2.0 / n_level
    return 1.5 * 1.7 * step_size


def gradient_threshold(N: int, side: float = 2.0) -> float:
    return 2.5 * side / N


def fill_grid(udf_func: Callable[[Tensor], Tensor], N: int, max_batch: int = 2 ** 16,
              with_grads: bool = True) -> Tuple[Tensor, Tensor, Dict]:
    """Returns (udf[N,N,N], grads[N,N,N,3], stats).  Axis 0 is x (slowest), flat index
    i*N*N + j*N + k, point (i,j,k) at coordinate axis_coords(N)[(i,j,k)]."""
    ax = axis_coords(N)
    udf = torch.zeros(N, N, N)
    stats: Dict = {"levels": grid_levels(N), "fwd_per_level": [], "grad": 0}
    active = None
    for li, n in enumerate(stats["levels"]):
        s = N // n
        lat = udf[::s, ::s, ::s]                       # lattice view of the dense grid
        if li == 0:
            act = torch.ones(n, n, n, dtype=torch.bool)
            new = act
        else:
            act = active
            seen = torch.zeros(n, n, n, dtype=torch.bool)
            seen[::2, ::2, ::2] = True                  # already queried one level up
            new = act & ~seen
        ijk = new.nonzero()
        pts = torch.stack([ax[ijk[:, 0] * s], ax[ijk[:, 1] * s], ax[ijk[:, 2] * s]], dim=1)
        stats["fwd_per_level"].append(int(pts.shape[0]))
        lat[new] = sample_udf(udf_func, pts, max_batch)
        if n < N:
            close = act & (lat.abs() < refine_threshold(n))
            far = (act & ~close).nonzero()
            blocks = udf.view(n, s, n, s, n, s)
            blocks[far[:, 0], :, far[:, 1], :, far[:, 2], :] = lat[far[:, 0], far[:, 1], far[:, 2]][:, None, None, None]
            active = close.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)
    grads = torch.zeros(N, N, N, 3)
    if with_grads:
        gi = (udf < gradient_threshold(N)).nonzero()
        stats["grad"] = int(gi.shape[0])
        if gi.shape[0]:
            pts = torch.stack([ax[gi[:, 0]], ax[gi[:, 1]], ax[gi[:, 2]]], dim=1)
            grads[gi[:, 0], gi[:, 1], gi[:, 2]] = sample_grads(udf_func, pts, max_batch)
    return udf, grads, stats


def fill_grid_dense(udf_func, N: int, max_dist: float = 0.1, max_batch: int = 2 ** 16,
                    coords_range=(-1.0, 1.0)) -> Tuple[Tensor, Tensor]:
    """Dense alternative (use_fast_grid_filler=False): every point forward, gradients where
    udf < max_dist - 1e-3 with chunks of max_batch // 4."""
    spacing = (coords_range[1] - coords_range[0]) / (N - 1)
    ax = (torch.arange(N, dtype=torch.float32) * spacing) + coords_range[0]
    coords = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(N ** 3, 3)
    udf = sample_udf(udf_func, coords, max_batch)
    grads = torch.zeros(N ** 3, 3)
    m = udf < (max_dist - 1e-3)
    if m.any():
        grads[m] = sample_grads(udf_func, coords[m], max_batch // 4)
    return udf.reshape(N, N, N), grads.reshape(N, N, N, 3)


def analytic_field(p: Tensor) -> Tensor:
    """Decoder-independent test field of SURVEY.md §8c G10: a hemisphere shell joined to a
    half torus tube, clipped at 0.1.  Written with elementwise IEEE ops only (no reductions),
    so a point's value does not depend on where it sits in the batch — the device path and the
    reference enumerate the same points in different orders."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    rho2 = x * x + y * y
    r = torch.sqrt(rho2 + z * z)
    rho = torch.sqrt(rho2)
    d_up = (r - 0.6).abs()
    t = rho - 0.6
    d_dn = torch.sqrt(t * t + z * z)
    d = torch.where(z >= 0, d_up, d_dn)
    return torch.clamp(d, max=0.1)

"""Drop-ins for the grid part of the reference's meshudf/meshudf.py, on libsurfd_hip.so.

  GridFiller(N).fill_grid(udf_func, max_batch)   <- meshudf.py:23-206
  sample_udf / sample_grads                       <- meshudf.py:209-251
  get_udf_and_grads                               <- meshudf.py:254-304 (dense variant)
  get_mesh_from_udf                               <- meshudf.py:307-514 (grids on the GPU, native marching cubes on
        the host, probe filter on the GPU, mesh cleaning / border smoothing in surfd_amd.meshproc)

Two execution modes, same algorithm and same device-side index kernels:
  * native  — ``udf_func`` was made by ``surfd_amd.cbndec.make_udf_func``: one C call fills
              udf[N,N,N] and grads[N,N,N,3] with no host round trip (surfd_grid_fill);
  * callback — any ``udf_func(Tensor[n,3]) -> Tensor[n]`` (the reference contract): per level
              the device emits the query points, the callable is evaluated in chunks of
              ``max_batch`` and the values are committed back; gradients by autograd exactly as
              the reference's sample_grads.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List,  Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import _native as N


def _chunks(n: int, size: int):
    for lo in range(0, n, size):
        yield lo, min(lo + size, n)


def sample_udf(udf_func: Callable[[Tensor], Tensor], coords: Tensor, max_batch: int, grad: bool = False) -> Tensor:
    """udf of `coords` [n,3] in chunks of `max_batch` (meshudf.py:209-228 behaviour); no autograd graph unless `grad`."""
    out = torch.zeros(coords.shape[0], device=coords.device)
    ctx = torch.enable_grad if grad else torch.no_grad
    for lo, hi in _chunks(coords.shape[0], max_batch):
        with ctx():
            out[lo:hi] = udf_func(coords[lo:hi])
    return out


def sample_grads(udf_func: Callable[[Tensor], Tensor], coords: Tensor, max_batch: int) -> Tensor:
    """-normalize(d udf / d p) per point (meshudf.py:231-251 behaviour).  Three routes: a field that knows how to
    shard itself (parallel.ShardedField), the native decoder (one fused forward+reverse kernel, no autograd), or an
    arbitrary callable differentiated with autograd chunk by chunk like the reference."""
    if hasattr(udf_func, "grads"):
        return udf_func.grads(coords, max_batch)
    native = getattr(udf_func, "_surfd_native", None)
    if native is not None:
        dec, lat, sample = native
        return dec.udf_and_ngrad(coords, dec._bind_single(lat) if sample is None else sample)[1]
    out = torch.zeros(coords.shape[0], 3, device=coords.device)
    for lo, hi in _chunks(coords.shape[0], max_batch):
        leaf = coords[lo:hi].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            (g,) = torch.autograd.grad(udf_func(leaf).sum(), leaf)
        out[lo:hi] = -F.normalize(g, dim=1)
    return out


class GridFiller:
    """Coarse-to-fine grid evaluation, 32^3 -> ... -> N^3 (see module docstring)."""

    def __init__(self, final_resolution: int, voxel_origin: Tuple[int, int, int] = (-1, -1, -1),
                 cube_side_length: float = 2.0):
        if tuple(voxel_origin) != (-1, -1, -1) or cube_side_length != 2.0:
            raise NotImplementedError("every reference call site uses the default [-1,1]^3 cube")
        self.N_max = final_resolution
        self.num_samples = final_resolution ** 3
        self.N_levels = [32 * (2 ** i) for i in range(int(math.log2(self.N_max) - 4))]
        self.voxel_origin = voxel_origin
        self.cube_side_length = cube_side_length
        self.voxel_size = cube_side_length / (self.N_max - 1)
        self._handle = None
        self.last_stats: Optional[Dict] = None

    def _native(self):
        L = N.lib()
        if self._handle is None:
            h = C.c_void_p()
            N.check(L.surfd_grid_create(self.N_max, C.byref(h)))
            # thresholds as Python computes them in the reference, rounded to fp32 the way
            # torch rounds a Python scalar meeting a float32 tensor
            refine = [float(torch.tensor(1.5 * 1.7 * (2.0 / n), dtype=torch.float32)) for n in self.N_levels]
            grad_thr = float(torch.tensor(2.5 * self.cube_side_length / self.N_max, dtype=torch.float32))
            voxel = float(torch.tensor(self.voxel_size, dtype=torch.float32))
            arr = (C.c_float * len(refine))(*refine)
            N.check(L.surfd_grid_set_thresholds(h, arr, len(refine), grad_thr, voxel, float(self.voxel_origin[0])))
            self._handle = h
        return L, self._handle

    def _stats(self) -> Dict:
        L, h = self._native()
        st = N.GridStats()
        N.check(L.surfd_grid_get_stats(h, C.byref(st), N.stream()))
        return {"levels": list(st.levels[:st.n_levels]), "fwd_per_level": list(st.fwd_points[:st.n_levels]),
                "grad": int(st.grad_points)}

    def totals(self, reset: bool = True) -> Dict:
        """Running totals of every fused fill on this filler since the last reset (kept on the device by the fills, one
        host read here): forward queries per level, forward+backward queries, fills."""
        L, h = self._native()
        st, n = N.GridStats(), C.c_int64()
        N.check(L.surfd_grid_get_totals(h, C.byref(st), C.byref(n), int(reset), N.stream()))
        return {"levels": list(st.levels[:st.n_levels]), "fwd_per_level": list(st.fwd_points[:st.n_levels]),
                "grad": int(st.grad_points), "fills": int(n.value)}

    def fill_grid(self, udf_func: Callable[[Tensor], Tensor], max_batch: int, with_grads: bool = True,
                  out: Optional[Tuple[Tensor, Optional[Tensor]]] = None, stats: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
        L, h = self._native()
        Nn = self.N_max
        dev = torch.device("cuda", torch.cuda.current_device())
        if out is None:
            udf = torch.empty(Nn, Nn, Nn, device=dev, dtype=torch.float32)
            grads = torch.empty(Nn, Nn, Nn, 3, device=dev, dtype=torch.float32) if with_grads else None
        else:
            udf, grads = out
        st = N.stream()
        native = getattr(udf_func, "_surfd_native", None)
        if native is not None:
            dec, lat, sample = native
            s = dec._bind_single(lat) if sample is None else sample
            _, dh = dec._native()
            N.check(L.surfd_grid_fill(h, dh, s, N.ptr(udf), N.ptr(grads), st))
        else:
            N.check(L.surfd_grid_begin(h, N.ptr(udf), N.ptr(grads), st))
            n = C.c_int64()
            for level in range(len(self.N_levels)):
                N.check(L.surfd_grid_level_points(h, level, None, 0, C.byref(n), st))
                xyz = torch.empty(max(n.value, 1), 3, device=dev, dtype=torch.float32)
                N.check(L.surfd_grid_level_points(h, level, N.ptr(xyz), n.value, C.byref(n), st))
                vals = sample_udf(udf_func, xyz[:n.value], max_batch).float().contiguous()
                N.check(L.surfd_grid_level_commit(h, level, N.ptr(vals), n.value, st))
            if with_grads:
                N.check(L.surfd_grid_grad_points(h, None, 0, C.byref(n), st))
                if n.value:
                    xyz = torch.empty(n.value, 3, device=dev, dtype=torch.float32)
                    N.check(L.surfd_grid_grad_points(h, N.ptr(xyz), n.value, C.byref(n), st))
                    ng = sample_grads(udf_func, xyz, max_batch).float().contiguous()
                    N.check(L.surfd_grid_grad_commit(h, N.ptr(ng), n.value, st))
        if stats:
            self.last_stats = self._stats()
        return udf, grads

    # ---- grid-shard mode -------------------------------------------------------------------------------------------------
    SHARD_DEFAULT_CAPACITY = 1 << 24          # points per level: the synthetic decoder's 8 %-occupancy field at 512^3 has 13.0 M on its last level
    SHARD_DEFAULT_GRAD_CAPACITY = 1 << 21     # gradient points (12 B each): 0.31 M for that field
    SHARD_FLOOR = 1 << 16

    def shard_capacities(self, capacity=None, grad_capacity=None) -> Tuple[List[int], int]:
        """Per-level exchange capacities (points) and the gradient capacity for fill_grid_sharded: an int bounds every level, a
        list names each level's own, None takes what plan_shard_capacities / the adaptive mode learned (the library defaults
        before that).  A level never needs more than its lattice (level 0) or 7 children per cell of the previous one."""
        nl = len(self.N_levels)
        upper = [self.N_levels[0] ** 3] + [7 * self.N_levels[l - 1] ** 3 for l in range(1, nl)]
        learned = getattr(self, "_shard_plan", None)
        if capacity is None:
            caps = list(learned[0]) if learned else [self.SHARD_DEFAULT_CAPACITY] * nl
        elif isinstance(capacity, (list, tuple)):
            if len(capacity) != nl:
                raise ValueError(f"capacity lists one entry per level ({nl}), got {len(capacity)}")
            caps = [int(c) for c in capacity]
        else:
            caps = [int(capacity)] * nl
        if grad_capacity is None:
            gcap = learned[1] if learned else self.SHARD_DEFAULT_GRAD_CAPACITY
        else:
            gcap = int(grad_capacity)
        return [max(1, min(c, u)) for c, u in zip(caps, upper)], max(1, min(gcap, self.N_max ** 3))

    def plan_shard_capacities(self, stats: Sequence[Dict], world: int = 1, margin: float = 1.5, grad_margin: float = 2.0) -> Tuple[List[int], int]:
        """Exchange capacities from measured fills (`stats`: GridFiller.last_stats of one or more shapes — every rank holds the
        same grids, hence the same counts, hence the same plan: the collectives' sizes agree by construction): per level
        `margin` x the largest count seen, rounded up to whole 64-point tiles of every rank, at least 2^16 points.  A thin-shell
        level then travels as a few MB instead of the 64 MB of the library default.  Remembered for capacity=None."""
        nl = len(self.N_levels)
        unit = 64 * max(1, world)
        up = lambda v: ((int(math.ceil(v)) + unit - 1) // unit) * unit          # noqa: E731
        caps = [up(max(self.SHARD_FLOOR, margin * max(st["fwd_per_level"][l] for st in stats))) for l in range(nl)]
        gcap = up(max(self.SHARD_FLOOR, grad_margin * max(st["grad"] for st in stats)))
        self._shard_plan = self.shard_capacities(caps, gcap)
        return self._shard_plan

    def shard_overflows(self, reset: bool = True) -> int:
        """Exchange buffers that were shorter than the list they carried, over the sharded fills since the last reset (kept on the
        device by the commits; one 8-byte read).  Non-zero: at least one grid of that span is incomplete."""
        L, h = self._native()
        n = C.c_int64()
        N.check(L.surfd_grid_shard_overflows(h, C.byref(n), int(reset), N.stream()))
        return int(n.value)

    def fill_grid_sharded(self, udf_func, rank: int = 0, world: int = 1, with_grads: bool = True,
                          out: Optional[Tuple[Tensor, Optional[Tensor]]] = None, stats: bool = True, capacity=None,
                          grad_capacity=None, exchange: Optional[Callable[[Tensor], None]] = None, simulate_ranks: bool = False,
                          adaptive: bool = False):
        """Grid-shard mode, native: ONE shape's grid evaluated by `world` ranks.  Every rank calls this with the
        same latent; rank r runs the native decoder on the 64-point tiles r, r + world, ... of each level's voxel-ordered point
        list, compacts them into its SEGMENT of capacity / world points (surfd_grid_shard_pack), `exchange(gathered, own)`
        all-gathers the ranks' segments (default: parallel.gather_segments — ncclAllGather over xGMI with backend "nccl";
        world = 1: nothing), and every rank commits the whole level from the gathered layout, derives the same next level and ends
        with the same grid.  Nothing is read back between levels: list lengths stay on the
        device, a level's exchange buffer holds `capacity` points (an int, one per level as a list, or None = the learned plan /
        the library default 2^24), the gradient buffer `grad_capacity` (12 B each; default 2^21).  A list longer than its buffer
        is cut — and counted on the device (shard_overflows()); with stats=True the counts are read back and a cut raises.
        adaptive=True (one read per shape): the capacities follow the field — every fill re-plans them from its own counts
        (plan_shard_capacities) and a fill that was cut is repeated with buffers sized from what it needed; identical on every
        rank because the counts are.  `self.shard_bytes_exchanged` sums the bytes a rank RECEIVES from the others
        ((world - 1) / world of every gathered buffer).
        simulate_ranks=True (tests on one device): this process plays every rank in turn — the sharding logic without a second GPU.
        The result equals the fused single-rank fill bit for bit."""
        native = getattr(udf_func, "_surfd_native", None)
        if native is None:
            raise RuntimeError("fill_grid_sharded needs a udf_func from surfd_amd.make_udf_func (the native decoder)")
        L, h = self._native()
        dec, lat, sample = native
        smp = dec._bind_single(lat) if sample is None else sample
        _, dh = dec._native()
        Nn = self.N_max
        dev = torch.device("cuda", torch.cuda.current_device())
        if out is None:
            udf = torch.empty(Nn, Nn, Nn, device=dev, dtype=torch.float32)
            grads = torch.empty(Nn, Nn, Nn, 3, device=dev, dtype=torch.float32) if with_grads else None
        else:
            udf, grads = out
        st = N.stream()
        if exchange is None and world > 1 and not simulate_ranks:
            from .parallel import gather_segments
            exchange = gather_segments
        ranks = range(world) if simulate_ranks else (rank,)
        if not hasattr(self, "shard_bytes_exchanged"):
            self.shard_bytes_exchanged = 0
        for attempt in range(3):
            caps, gcap = self.shard_capacities(capacity, grad_capacity)
            if world > 1:                         # a rank's segment is whole 64-point tiles: capacity / world
                unit = 64 * world
                caps = [((c + unit - 1) // unit) * unit for c in caps]
                gcap = ((gcap + unit - 1) // unit) * unit
            need = max(max(caps), 3 * gcap if grads is not None else 0)
            buf = getattr(self, "_shard_buf", None)
            if buf is None or buf.numel() < need or buf.device != dev:
                buf = self._shard_buf = torch.empty(need, device=dev, dtype=torch.float32)
            gat = own = None
            if world > 1:                         # the gathered layout [world][segment] and this rank's segment
                gat = getattr(self, "_shard_gat", None)
                if gat is None or gat.numel() < need or gat.device != dev:
                    gat = self._shard_gat = torch.empty(need, device=dev, dtype=torch.float32)
                own = getattr(self, "_shard_own", None)
                if own is None or own.numel() < need // world or own.device != dev:
                    own = self._shard_own = torch.empty(need // world, device=dev, dtype=torch.float32)
            N.check(L.surfd_grid_shard_begin(h, N.ptr(udf), N.ptr(grads), st))

            def exchange_step(level, vals, cap, width):
                """this process's rank(s): tiles -> segment(s) -> the gathered buffer every rank commits from"""
                seg = cap // world * width
                for r in ranks:
                    dst = gat[r * seg:(r + 1) * seg] if simulate_ranks else own[:seg]
                    N.check(L.surfd_grid_shard_pack(h, level, r, world, N.ptr(vals), cap, N.ptr(dst), st))
                if exchange is not None:
                    exchange(gat[:world * seg], own[:seg])
                self.shard_bytes_exchanged += 4 * seg * (world - 1)      # received from the other ranks (also when one process plays them)
                return gat[:world * seg]

            nl = len(caps)
            for level, cap in enumerate(caps):
                vals = buf[:cap]
                for r in ranks:
                    N.check(L.surfd_grid_shard_level_eval(h, dh, smp, level, r, world, N.ptr(vals), cap, st))
                src = exchange_step(level, vals, cap, 1) if world > 1 else vals
                N.check(L.surfd_grid_shard_level_commit(h, level, N.ptr(src), cap, world, st))
            if grads is not None:
                ng = buf[:3 * gcap]
                for r in ranks:
                    N.check(L.surfd_grid_shard_grad_eval(h, dh, smp, r, world, N.ptr(ng), gcap, st))
                src = exchange_step(nl, ng, gcap, 3) if world > 1 else ng
                N.check(L.surfd_grid_shard_grad_commit(h, N.ptr(src), gcap, world, st))
            if not (stats or adaptive):
                break
            self.last_stats = self._stats()
            over = [(l, c, cap) for l, (c, cap) in enumerate(zip(self.last_stats["fwd_per_level"], caps)) if c > cap]
            cut = bool(over) or (grads is not None and self.last_stats["grad"] > gcap)
            if adaptive:
                # plan from the largest counts seen on this filler so far (fields differ from shape to shape: following only the
                # last one would repeat every shape that is larger than its predecessor); a cut level hides how long its children
                # would have been: twice the margin, and go again
                seen = getattr(self, "_shard_seen", None)
                cur = {"fwd_per_level": list(self.last_stats["fwd_per_level"]), "grad": self.last_stats["grad"]}
                if seen is not None:
                    cur = {"fwd_per_level": [max(a, b) for a, b in zip(cur["fwd_per_level"], seen["fwd_per_level"])], "grad": max(cur["grad"], seen["grad"])}
                self._shard_seen = cur
                self.plan_shard_capacities([cur], world, margin=3.0 if cut else 1.5, grad_margin=4.0 if cut else 2.0)
                if cut and attempt < 2:
                    self.shard_overflows(reset=True)
                    capacity = grad_capacity = None
                    continue
            if cut:
                raise RuntimeError(f"fill_grid_sharded: exchange capacity too small for this field: per level {caps}, gradient points {gcap} "
                                   f"(levels over: {over}, gradient points {self.last_stats['grad']} of {gcap}): pass larger ones, or adaptive=True")
            break
        return udf, grads

    def fill_grid_dense(self, udf_func, max_dist: float = 0.1, with_grads: bool = True):
        """get_udf_and_grads semantics on the native decoder (all N^3 points)."""
        native = getattr(udf_func, "_surfd_native", None)
        if native is None:
            raise RuntimeError("fill_grid_dense needs a udf_func from surfd_amd.make_udf_func")
        L, h = self._native()
        dec, lat, sample = native
        s = dec._bind_single(lat) if sample is None else sample
        _, dh = dec._native()
        Nn = self.N_max
        dev = torch.device("cuda", torch.cuda.current_device())
        udf = torch.empty(Nn, Nn, Nn, device=dev, dtype=torch.float32)
        grads = torch.empty(Nn, Nn, Nn, 3, device=dev, dtype=torch.float32) if with_grads else None
        thr = float(torch.tensor(max_dist - 1e-3, dtype=torch.float32))
        N.check(L.surfd_grid_fill_dense(h, dh, s, thr, N.ptr(udf), N.ptr(grads), N.stream()))
        self.last_stats = self._stats()
        return udf, grads

    def __del__(self):
        try:
            if self._handle is not None:
                N.lib().surfd_grid_destroy(self._handle)
        except Exception:
            pass


def fill_grids(fillers: Sequence["GridFiller"], decoder, samples: Sequence[int], outs: Sequence[Tuple[Tensor, Optional[Tensor]]]) -> None:
    """Coarse-to-fine grids of up to 8 shapes TOGETHER (throughput form of calling GridFiller.fill_grid shape after
    shape, as sample/generate_*.py do): `samples[i]` is the index of shape i's latent among the ones bound with
    ``decoder.bind_latents``, `fillers[i]` its own GridFiller, `outs[i]` = (udf [N,N,N], grads [N,N,N,3] | None).  Every
    refinement level of all shapes is one launch of the persistent decoder kernel; the values are bit-identical to the
    per-shape calls."""
    n = len(fillers)
    if not (1 <= n <= 8) or len(samples) != n or len(outs) != n:
        raise ValueError("fill_grids: 1..8 shapes, one filler / sample index / output pair each")
    L = N.lib()
    _, dh = decoder._native()
    hs = (C.c_void_p * n)(*[f._native()[1] for f in fillers])
    ss = (C.c_int * n)(*[int(s) for s in samples])
    us = (C.c_void_p * n)(*[o[0].data_ptr() for o in outs])
    gs = (C.c_void_p * n)(*[(o[1].data_ptr() if o[1] is not None else None) for o in outs])
    N.check(L.surfd_grid_fill_batch(hs, n, dh, ss, us, gs, N.stream()))


def get_udf_and_grads(udf_func, coords_range: Tuple[float, float], max_dist: float, N: int, max_batch: int):
    if tuple(coords_range) != (-1, 1):
        raise NotImplementedError("every reference call site uses coords_range=(-1, 1)")
    return GridFiller(N).fill_grid_dense(udf_func, max_dist)


def get_mesh_from_udf(udf_func: Callable[[Tensor], Tensor], coords_range: Tuple[float, float], max_dist: float,
                      N: int = 128, smooth_borders: bool = True, differentiable: bool = True,
                      max_batch: int = 2 ** 12, use_fast_grid_filler: bool = True) -> Tuple[Tensor, Tensor]:
    """Triangulated mesh of the zero set of a UDF — same signature, defaults and return value as the reference
    (meshudf/meshudf.py:307-514): (vertices [V,3] float32, faces [F,3] int64) on the device.

    Stages: coarse-to-fine grid + gradients on the GPU (one C call with the native decoder) -> native UDF marching
    cubes on the host (csrc/mcubes.cpp, bit-identical to the reference's Cython extension) -> faces with any edge
    end/mid point farther than 1/N from the surface are dropped (3 x 3F extra decoder queries, on the GPU) -> mesh
    cleaning and border smoothing (surfd_amd.meshproc: numpy restatements of the trimesh calls) -> optionally the
    reference's re-attachment of gradients to the vertex positions (``differentiable=True``; the sample scripts pass
    False)."""
    from . import meshproc as mp
    from .mcubes import udf_mc_lewiner
    if not use_fast_grid_filler:
        udf, gradients = get_udf_and_grads(udf_func, coords_range, max_dist, N, max_batch)
    else:
        udf, gradients = GridFiller(N).fill_grid(udf_func, max_batch)
    udf[udf < 0] = 0
    N = udf.shape[0]
    dev = udf.device
    spacing = (coords_range[1] - coords_range[0]) / (N - 1)
    vertices, faces, _, _ = udf_mc_lewiner(udf.detach().cpu().numpy(), gradients.detach().cpu().numpy(), spacing=[spacing] * 3)
    del udf, gradients
    vertices = vertices + coords_range[0]
    faces = faces.astype(np.int64)

    # faces whose corners or edge midpoints sit at large udf values are artefacts: drop them (meshudf.py:354-378)
    edges, edge_face = mp.edges_of_faces(faces)
    pa, pb = vertices[edges[:, 0]], vertices[edges[:, 1]]
    probes = torch.from_numpy(np.vstack((pa, pb, (pa + pb) / 2))).float().to(dev)
    far = sample_udf(udf_func, probes, max_batch).cpu().numpy() > 1 / N
    keep = np.ones(len(faces), dtype=bool)
    keep[np.unique(np.tile(edge_face, 3)[far])] = False
    vertices, faces = mp.clean_until_stable(vertices, faces[keep])
    if smooth_borders:
        vertices = mp.smooth_borders(vertices, faces, lam=0.3, iterations=20)
    final_verts = torch.tensor(vertices).float().to(dev)
    final_faces = torch.tensor(faces).long().to(dev)
    if not differentiable:
        return final_verts, final_faces
    return _reattach_gradients(udf_func, vertices, faces, final_verts, final_faces, 1 / N, max_batch, mp)


def get_watertight_mesh(udf_func: Callable[[Tensor], Tensor], N: int, max_batch: int = 2 ** 16, level: float = 0.01,
                        normalize: bool = False, classic: bool = True):
    """The --watertight path of the reference's text / image drivers (sample/generate_text.py:132-158 with the
    CPU-resident GridFiller of utils/utils.py:151-339): coarse-to-fine UDF grid WITHOUT gradients, then the closed
    `level` iso-surface of the UDF (a thin solid around the sheet).  Returns (vertices [V,3] float64, faces [F,3]) on
    the host.  Vertices are in voxel-index units like the file the reference ends up writing (it normalises a
    component it then does not export); ``normalize=True`` maps them to [-1, 1]^3.  PyMCubes is not available here:
    the surface comes from the native level-set mesher (classic triangle table by default) — same level set,
    triangulation parity with PyMCubes unpinned.  Small-component removal is the caller's next step (5000 faces)."""
    from .mcubes import marching_cubes
    udf, _ = GridFiller(N).fill_grid(udf_func, max_batch, with_grads=False)
    udf[udf < 0] = 0
    verts, faces = marching_cubes(udf.detach().cpu().numpy(), level, classic=classic)
    if normalize:
        verts = verts * (2.0 / N) - 1.0
    return verts, faces


def _reattach_gradients(udf_func, vertices, faces, verts, final_faces, th_dist, max_batch, mp):
    """meshudf.py:439-512: the vertex positions are re-expressed through udf samples taken at +-th_dist along the
    normals (and, on borders, along the outward in-surface direction) so that d(vertices)/d(udf parameters) exists;
    numerically the positions do not move (every added term is x - x.detach())."""
    dev = verts.device
    normals = torch.tensor(mp.vertex_normals_by_angle(vertices, faces)).float().to(dev)
    s1 = sample_udf(udf_func, verts + th_dist * normals, max_batch, True).unsqueeze(-1)
    s2 = sample_udf(udf_func, verts - th_dist * normals, max_batch, True).unsqueeze(-1)
    z = th_dist * s1 * normals - th_dist * s2 * normals
    new_verts = verts - z + z.detach()
    rows = mp.border_edge_rows(faces)
    if len(rows):
        e = np.sort(mp.edges_of_faces(faces)[0][rows], axis=1)
        partner = {}
        for u, v in e:                                   # every border vertex keeps ONE of its border edges (the last seen)
            partner[int(u)] = int(v)
            partner[int(v)] = int(u)
        u_b = np.fromiter(partner.keys(), dtype=np.int64)
        v_b = np.fromiter(partner.values(), dtype=np.int64)
        edge = torch.tensor(vertices[v_b] - vertices[u_b]).float().to(dev)
        out_vec = torch.cross(edge, normals[u_b], dim=1)
        out_vec = out_vec / (torch.norm(out_vec, dim=1, keepdim=True) + 1e-6)
        bverts = torch.tensor(vertices[u_b]).float().to(dev)
        b1 = sample_udf(udf_func, bverts + 3 * th_dist * out_vec, max_batch, True).unsqueeze(-1)
        b2 = sample_udf(udf_func, bverts - 3 * th_dist * out_vec, max_batch, True).unsqueeze(-1)
        out_vec = (-torch.argmax(torch.stack((b1, b2)), dim=0) * 2 + 1) * out_vec          # towards the larger udf
        real = (b1 + b2)[:, 0] > th_dist                   # a true border leaves the surface on at least one side
        big = torch.max(b1, b2)[real.unsqueeze(-1)]
        shift = (th_dist * (big - big.detach())).unsqueeze(-1)
        idx = torch.from_numpy(u_b).to(dev)[real]
        new_verts[idx] = new_verts[idx] - shift * out_vec[real]
    return new_verts, final_faces

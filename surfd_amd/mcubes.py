"""UDF marching cubes on the host through libsurfd_hip.so (csrc/mcubes.cpp).

  udf_mc_lewiner(volume, grads, spacing, ...)   <- meshudf/_marching_cubes_lewiner.py:87-154 (same name, arguments, returns)

The C++ mesher restates the reference's Cython extension (meshudf/_marching_cubes_lewiner_cy.pyx) and reproduces its
output bit for bit (tests/test_mcubes_cpu.py compares with the reference's own compiled extension and with committed
fixtures).  Host-side code: no GPU involved; one call meshes one shape on one core.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np

from . import _native as N


def _run(volume: np.ndarray, grads, step_size: int, level: float = 0.0, classic: bool = False):
    L = N.lib()
    nz, ny, nx = volume.shape
    h = C.c_void_p()
    if grads is None:
        N.check(L.surfd_mc_iso(volume.ctypes.data_as(C.c_void_p), nz, ny, nx, float(level), int(classic), step_size, C.byref(h)))
    else:
        N.check(L.surfd_mc_udf(volume.ctypes.data_as(C.c_void_p), grads.ctypes.data_as(C.c_void_p), nz, ny, nx, step_size, C.byref(h)))
    try:
        nv, nf = L.surfd_mc_num_vertices(h), L.surfd_mc_num_faces(h)
        verts = np.empty((nv, 3), np.float32)
        normals = np.empty((nv, 3), np.float32)
        values = np.empty((nv,), np.float32)
        faces = np.empty((nf, 3), np.int32)
        N.check(L.surfd_mc_copy(h, verts.ctypes.data_as(C.c_void_p), faces.ctypes.data_as(C.c_void_p),
                                normals.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p)))
    finally:
        L.surfd_mc_destroy(h)
    return verts, faces, normals, values


def udf_mc_lewiner(volume, grads, spacing=(1.0, 1.0, 1.0), gradient_direction="descent", step_size=1,
                   allow_degenerate=True, use_classic=False, mask=None) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """(vertices[V,3] in (z,y,x)*spacing, faces[F,3] int32, normals[V,3], values[V]) of the zero set of the UDF
    `volume[N,N,N]`, the signs taken from the gradient field `grads[N,N,N,3]`."""
    if not isinstance(volume, np.ndarray) or volume.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    if len(spacing) != 3:
        raise ValueError("`spacing` must consist of three floats.")
    step_size = int(step_size)
    if step_size < 1:
        raise ValueError("step_size must be at least one.")
    if use_classic or mask is not None or not allow_degenerate:
        raise NotImplementedError("the sampling path calls udf_mc_lewiner with its defaults only (meshudf.py:347)")
    if gradient_direction not in ("descent", "ascent"):
        raise ValueError(f"Incorrect input {gradient_direction} in `gradient_direction`, see docstring.")
    volume = np.ascontiguousarray(volume, np.float32)
    grads = np.ascontiguousarray(grads, np.float32)
    if grads.shape != volume.shape + (3,):
        raise ValueError("grads must be volume.shape + (3,)")
    verts, faces, normals, values = _run(volume, grads, step_size)
    if not len(verts):
        raise RuntimeError("No surface found at the given iso value.")
    if gradient_direction == "ascent":
        faces = np.fliplr(faces)
    if not np.array_equal(spacing, (1, 1, 1)):
        verts = verts * np.r_[spacing]            # float64, as the reference's multiplication by np.r_[spacing]
    return verts, faces, normals, values


def marching_cubes(volume, isovalue: float, classic: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """Level-set marching cubes with PyMCubes' call shape (``mcubes.marching_cubes(u, 0.01)``, generate_text.py:139-141):
    (vertices [V,3] float64 in voxel-index units along the volume's axes, faces [F,3]).  ``classic=True`` uses the
    original triangle table like PyMCubes; the triangulation / vertex order of PyMCubes itself is not reproduced
    (package absent here: parity unpinned), the surface is the same linear-interpolation level set."""
    volume = np.ascontiguousarray(volume, np.float32)
    if volume.ndim != 3 or min(volume.shape) < 2:
        raise ValueError("volume must be a 3D array of at least 2x2x2")
    v, f, _, _ = _run(volume, None, 1, isovalue, classic)
    return v.astype(np.float64), f.astype(np.int64)          # wound outward for the solid {volume < isovalue}


def lut_tables() -> dict:
    """The marching-cubes case tables the library was built with, as int8 arrays (tests feed them to the reference's
    compiled extension on hosts without the reference tree)."""
    L = N.lib()
    out = {}
    for i in range(L.surfd_mc_lut_count()):
        name, vals, ndim = C.c_char_p(), C.POINTER(C.c_byte)(), C.c_int()
        dims = (C.c_int * 3)()
        N.check(L.surfd_mc_lut(i, C.byref(name), C.byref(vals), C.byref(ndim), dims))
        shape = tuple(dims[k] for k in range(ndim.value))
        n = int(np.prod(shape))
        out[name.value.decode()] = np.ctypeslib.as_array(vals, shape=(n,)).astype(np.int8).reshape(shape).copy()
    return out


def bench_e2(resolution: int, e1_shapes_per_s: float, threads: int = 8) -> dict:
    """End point E2 of SURVEY.md §8d for bench.py: E1 + device->host copy of (udf, grads) + marching cubes.  The host
    stage is measured on one thin-shell field of the benchmark's resolution (the occupancy of a trained model:
    222 793 vertices at 512^3): seconds per shape on one core, and the rate of `threads` independent shapes meshed
    concurrently (one per host core; ctypes releases the GIL).  In a pipelined service the host stage overlaps the GPU
    work on later batches.  These are STAGE figures for the line of an E1 run; the end-to-end rate through the meshes is
    measured, not composed: `bench.py --endpoint e2` (BandMesher below)."""
    import os
    import time
    from concurrent.futures import ThreadPoolExecutor

    import torch

    from .meshudf import GridFiller

    def field(c):
        x, y, z = c[:, 0], c[:, 1], c[:, 2]
        up = (torch.sqrt(x * x + y * y + z * z) - 0.6).abs()
        rho = torch.sqrt(x * x + y * y) - 0.6
        return torch.clamp(torch.where(z >= 0, up, torch.sqrt(rho * rho + z * z)), max=0.1)

    class Field:
        def __call__(self, c):
            return field(c)

        def grads(self, c, max_batch):
            p = c.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                (g,) = torch.autograd.grad(field(p).sum(), p)
            return -torch.nn.functional.normalize(g, dim=1)

    udf, grads = GridFiller(resolution).fill_grid(Field(), 2 ** 30)
    udf[udf < 0] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u, g = udf.cpu().numpy(), grads.cpu().numpy()
    d2h_s = time.perf_counter() - t0
    del udf, grads
    torch.cuda.empty_cache()
    t0 = time.perf_counter()
    v, f, _, _ = _run(u, g, 1)
    one_s = time.perf_counter() - t0
    threads = max(1, min(threads, os.cpu_count() or 1))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda _: _run(u, g, 1), range(threads)))
    par_rate = threads / (time.perf_counter() - t0)
    return {"status": "stage figures, measured on the side of an E1 run — NOT an end-to-end rate; the timed end point is `bench.py --endpoint e2` "
                      "(every mesh finished inside the timed region)",
            "field": f"thin shell, {resolution}^3 ({len(v)} vertices / {len(f)} faces)",
            "dense_d2h_s_per_shape": d2h_s, "mc_s_per_shape_one_core_dense_input": one_s, "mc_threads": threads,
            "mc_shapes_per_s": par_rate, "e1_shapes_per_s": e1_shapes_per_s,
            "note": "pageable whole-volume D2H (what the reference does, meshudf.py:344-349; the timed path copies the near-surface band "
                    "instead) + native marching cubes, one shape per core; reference marching cubes: 10.6 s per shape at 512^3 on one core (BASELINE.md)"}


# ---- sparse hand-off: the mesher on the near-surface band only (csrc/mcubes.cpp: surfd_mc_udf_band) -------------------------
def band_threshold(n: int) -> float:
    """float32(1.74 * 2 / (n - 1)): the largest corner value a cube may have for the mesher to look at it
    (_marching_cubes_lewiner_cy.pyx:1131,1157-1158), exactly as the library compares."""
    t = C.c_float()
    N.check(N.lib().surfd_mc_band_threshold(int(n), C.byref(t)))
    return float(t.value)


class McScratch:
    """Host scratch volume of one meshing thread (n^3 voxels: value, gradient, state byte; pages appear where bands touch
    them): surfd_mc_udf_band scatters a shape's band in, meshes, takes it out again."""

    def __init__(self, n: int):
        self.n = int(n)
        h = C.c_void_p()
        N.check(N.lib().surfd_mc_scratch_create(self.n, C.byref(h)))
        self._h = h

    def mesh(self, index, packed, count: int, step_size: int = 1):
        """index: int32 voxel indices (pointer or array), packed: [count, 4] float32 (udf, gx, gy, gz) ->
        (vertices, faces, normals, values) exactly as udf_mc_lewiner(...) with spacing 1 on the dense volumes."""
        L = N.lib()
        if isinstance(index, np.ndarray):
            index = np.ascontiguousarray(index, np.int32)
            packed = np.ascontiguousarray(packed, np.float32)
            ip, pp = index.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)
        else:
            ip, pp = index, packed
        h = C.c_void_p()
        N.check(L.surfd_mc_udf_band(self._h, ip, pp, int(count), int(step_size), C.byref(h)))
        try:
            nv, nf = L.surfd_mc_num_vertices(h), L.surfd_mc_num_faces(h)
            verts, normals = np.empty((nv, 3), np.float32), np.empty((nv, 3), np.float32)
            values, faces = np.empty((nv,), np.float32), np.empty((nf, 3), np.int32)
            N.check(L.surfd_mc_copy(h, verts.ctypes.data_as(C.c_void_p), faces.ctypes.data_as(C.c_void_p),
                                    normals.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p)))
        finally:
            L.surfd_mc_destroy(h)
        return verts, faces, normals, values

    def __del__(self):
        try:
            if self._h is not None:
                N.lib().surfd_mc_scratch_destroy(self._h)
        except Exception:
            pass


def band_of(volume: np.ndarray, grads: np.ndarray):
    """numpy restatement of the device compaction (surfd_band_compact): (index int32 [n], packed float32 [n, 4]) of the voxels
    with udf <= band_threshold, in voxel order."""
    v = np.ascontiguousarray(volume, np.float32).reshape(-1)
    idx = np.nonzero(~(v > np.float32(band_threshold(volume.shape[0]))))[0].astype(np.int32)
    packed = np.empty((len(idx), 4), np.float32)
    packed[:, 0] = np.maximum(v[idx], 0)
    packed[:, 1:] = np.ascontiguousarray(grads, np.float32).reshape(-1, 3)[idx]
    return idx, packed


class BandMesher:
    """End point E2 executed: for every finished grid the device compacts the near-surface band
    (`submit`, stream-ordered on the caller's stream — no host sync), a pool of host threads fetches it over their own copy
    streams into pinned memory and runs the native UDF marching cubes on it, while the GPU goes on with the next shapes.
    `slots` band handles bound how far the GPU may run ahead of the mesher (submit blocks when all are in flight); `drain`
    returns when every submitted shape is meshed.  Meshes are what udf_mc_lewiner gives on the dense grids, bit for bit
    (tests/test_gpu_decoder_grid.py).  keep=True keeps (tag, vertices, faces) in .meshes, else only counts."""

    def __init__(self, n: int, threads: int = 8, slots: int = 16, capacity: int = 1 << 21, keep: bool = False):
        import queue
        import threading

        import torch
        self.n, self.keep = int(n), keep
        self.max_thr = band_threshold(n)
        capacity = max(1, min(int(capacity), self.n ** 3))     # surfd_band_create rejects more than N^3 (resolutions below 128)
        self.capacity = capacity
        L = N.lib()
        self._free, self._work = queue.Queue(), queue.Queue()
        self._handles = []
        for _ in range(slots):
            h = C.c_void_p()
            N.check(L.surfd_band_create(self.n, int(capacity), C.byref(h)))
            self._handles.append(h)
            self._free.put(h)
        self.meshes, self.errors = [], []
        self._lock = threading.Lock()
        self.reset_stats()
        self._threads_n = max(1, int(threads))
        dev = torch.cuda.current_device()

        def worker():
            torch.cuda.set_device(dev)
            scratch = McScratch(self.n)
            copy_stream = torch.cuda.Stream()
            while True:
                item = self._work.get()
                if item is None:
                    self._work.task_done()
                    return
                h, tag = item
                try:
                    t0 = time.perf_counter()
                    cnt, ip, pp = C.c_int64(), C.c_void_p(), C.c_void_p()
                    N.check(L.surfd_band_fetch(h, C.c_void_p(copy_stream.cuda_stream), C.byref(cnt), C.byref(ip), C.byref(pp)))
                    t1 = time.perf_counter()
                    v, f, _, _ = scratch.mesh(ip, pp, cnt.value)
                    t2 = time.perf_counter()
                    with self._lock:
                        self._st["shapes"] += 1; self._st["band_voxels"] += cnt.value
                        self._st["wait_and_copy_s"] += t1 - t0; self._st["mesh_s"] += t2 - t1
                        self._st["vertices"] += len(v); self._st["faces"] += len(f)
                        if self.keep:
                            self.meshes.append((tag, v, f))
                except BaseException as e:
                    self.errors.append(e)
                finally:
                    self._free.put(h)
                    self._work.task_done()

        import time
        self._threads = [threading.Thread(target=worker, daemon=True) for _ in range(self._threads_n)]
        for t in self._threads:
            t.start()

    def reset_stats(self):
        self._st = {"shapes": 0, "band_voxels": 0, "wait_and_copy_s": 0.0, "mesh_s": 0.0, "vertices": 0, "faces": 0}

    def submit(self, udf, grads, tag=None) -> None:
        if self.errors:                                        # e.g. a band larger than `capacity`, seen by a worker at fetch time:
            raise self.errors[0]                               # fail at the next submit, not after the whole run
        h = self._free.get()                                   # back-pressure: at most `slots` shapes between GPU and mesher
        N.check(N.lib().surfd_band_compact(h, N.ptr(udf), N.ptr(grads), C.c_float(self.max_thr), N.stream()))
        self._work.put((h, tag))

    def drain(self) -> None:
        self._work.join()
        if self.errors:
            raise self.errors[0]

    def stats(self) -> dict:
        s = dict(self._st)
        n = max(1, s["shapes"])
        return {"threads": self._threads_n, "slots": len(self._handles), "meshed_shapes": s["shapes"],
                "band_voxels_per_shape": s["band_voxels"] / n, "d2h_bytes_per_shape": 20.0 * s["band_voxels"] / n + 4,
                "dense_d2h_bytes_per_shape": 16.0 * self.n ** 3,
                "mesh_s_per_shape_one_thread": s["mesh_s"] / n, "wait_and_copy_s_per_shape": s["wait_and_copy_s"] / n,
                "vertices_per_shape": s["vertices"] / n, "faces_per_shape": s["faces"] / n}

    def close(self) -> None:
        for _ in self._threads:
            self._work.put(None)
        for t in self._threads:
            t.join()
        self._threads = []
        for h in self._handles:
            N.lib().surfd_band_destroy(h)
        self._handles = []

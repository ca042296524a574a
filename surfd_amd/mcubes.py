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
    work on later batches, so E2 = min(E1, host rate)."""
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
    return {"status": "measured", "field": f"thin shell, {resolution}^3 ({len(v)} vertices / {len(f)} faces)",
            "d2h_s_per_shape": d2h_s, "mc_s_per_shape_one_core": one_s, "mc_threads": threads,
            "mc_shapes_per_s": par_rate, "e1_shapes_per_s": e1_shapes_per_s,
            "value": min(e1_shapes_per_s, par_rate), "unit": "shapes/s",
            "note": "host stage (pageable D2H + native marching cubes, one shape per core) overlapped with the GPU work of later "
                    "batches; reference marching cubes: 10.6 s per shape at 512^3 on one core (BASELINE.md)"}

"""Synthetic UDF grids for the marching-cubes parity tests (deterministic; shared by the golden generator and the tests)."""
import numpy as np


def _grid(N):
    ax = (np.arange(N, dtype=np.float32) * np.float32(2.0 / (N - 1)) + np.float32(-1.0)).astype(np.float32)
    return np.meshgrid(ax, ax, ax, indexing="ij")


def _finish(d, gx, gy, gz, max_dist=0.1, grad_below=None, N=None):
    """udf = min(d, max_dist); grads = -normalize(grad d) where udf < 2.5 * 2 / N (GridFiller's band), else 0."""
    N = d.shape[0]
    udf = np.minimum(d, max_dist).astype(np.float32)
    g = np.stack([gx, gy, gz], -1).astype(np.float32)
    nrm = np.maximum(np.linalg.norm(g, axis=-1, keepdims=True), 1e-12).astype(np.float32)
    g = (-(g / nrm)).astype(np.float32)
    band = udf < (np.float32(2.5 * 2.0 / N) if grad_below is None else grad_below)
    g[~band] = 0
    return np.ascontiguousarray(udf), np.ascontiguousarray(g)


def two_spheres(N):
    x, y, z = _grid(N)
    def sph(cx, cy, cz, r):
        dx, dy, dz = x - cx, y - cy, z - cz
        rr = np.sqrt(dx * dx + dy * dy + dz * dz)
        s = rr - r
        sg = np.sign(s)
        return np.abs(s), sg * dx / np.maximum(rr, 1e-9), sg * dy / np.maximum(rr, 1e-9), sg * dz / np.maximum(rr, 1e-9)
    a, b = sph(-0.2, 0.0, 0.05, 0.45), sph(0.25, 0.1, -0.05, 0.4)
    pick = a[0] <= b[0]
    return _finish(np.where(pick, a[0], b[0]), np.where(pick, a[1], b[1]), np.where(pick, a[2], b[2]), np.where(pick, a[3], b[3]))


def open_sheet(N):
    """An open surface (a wavy sheet clipped to a disc): borders, the case UDFs exist for."""
    x, y, z = _grid(N)
    h = 0.15 * np.sin(3.0 * x) * np.cos(2.5 * y)
    s = z - h
    rho = np.sqrt(x * x + y * y)
    out = np.maximum(rho - 0.7, 0.0)
    d = np.sqrt(s * s + out * out)
    sg = np.sign(s)
    gx = sg * (-0.45 * np.cos(3.0 * x) * np.cos(2.5 * y)) + out * x / np.maximum(rho, 1e-9)
    gy = sg * (0.375 * np.sin(3.0 * x) * np.sin(2.5 * y)) + out * y / np.maximum(rho, 1e-9)
    gz = sg * np.ones_like(z)
    return _finish(d, gx, gy, gz)


def noisy_blob(N, seed=5):
    """|f| of a random smooth field with rotated (noisy) gradients and exact zeros: ambiguous cases, unsure votes."""
    rng = np.random.default_rng(seed)
    x, y, z = _grid(N)
    f = np.zeros_like(x); gx = np.zeros_like(x); gy = np.zeros_like(x); gz = np.zeros_like(x)
    for _ in range(6):
        k = rng.normal(size=3) * 3.0
        ph = rng.uniform(0, 6.28)
        a = rng.uniform(0.5, 1.0)
        arg = (k[0] * x + k[1] * y + k[2] * z + ph).astype(np.float32)
        f += a * np.sin(arg); c = a * np.cos(arg)
        gx += c * k[0]; gy += c * k[1]; gz += c * k[2]
    f = (f * 0.05).astype(np.float32)
    sg = np.sign(f)
    noise = rng.normal(scale=0.35, size=(3,) + x.shape).astype(np.float32)
    d = np.abs(f)
    d[d < 2e-4] = 0.0                                   # exact zeros are special in the voting (pyx:1261, 1289)
    return _finish(d, sg * gx * 0.05 + noise[0] * 0.1, sg * gy * 0.05 + noise[1] * 0.1, sg * gz * 0.05 + noise[2] * 0.1)


def thin_shell(N):
    """sphere shell above z = 0, rim below — the grid GridFiller produces for it."""
    import torch
    from oracle.gridfiller import analytic_field, fill_grid
    udf, grads, _ = fill_grid(analytic_field, N, max_batch=2 ** 30)
    u = udf.numpy().copy()
    u[u < 0] = 0
    return np.ascontiguousarray(u), np.ascontiguousarray(grads.numpy())


FIELDS = {"two_spheres": two_spheres, "open_sheet": open_sheet, "noisy_blob": noisy_blob, "thin_shell": thin_shell}

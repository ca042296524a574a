"""ORACLE (test infrastructure, never shipped as product): CPU restatement of the Surf-D
reverse-diffusion sampler.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Pinned by tests/golden/ (made by tools/make_golden.py from
the imported reference).

Reference anchors (relative to /root/reference):
  cosine schedule            diffusion/gaussian_diffusion.py:23-67
  derived float64 arrays     diffusion/gaussian_diffusion.py:144-180
  timestep respacing         diffusion/respace.py:7-60, 63-85, 123-128
  p_mean_variance (START_X, FIXED_SMALL)   gaussian_diffusion.py:258-363
  p_sample                   gaussian_diffusion.py:471-520
  p_sample_loop_progressive  gaussian_diffusion.py:635-708
  ddim_sample                gaussian_diffusion.py:711-761
  _extract_into_tensor       gaussian_diffusion.py:1329-1342 (float64 -> float32 cast)
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

Tensor = torch.Tensor


def cosine_betas(T: int = 1000, max_beta: float = 0.999) -> np.ndarray:
    def abar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - abar((i + 1) / T) / abar(i / T), max_beta) for i in range(T)], dtype=np.float64)


def linear_betas(T: int = 1000) -> np.ndarray:
    scale = 1000 / T
    return np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)


def ddim_timesteps(T: int, count: int) -> List[int]:
    for stride in range(1, T):
        if len(range(0, T, stride)) == count:
            return list(range(0, T, stride))
    raise ValueError(f"cannot create exactly {count} steps with an integer stride")


def space_timesteps(T: int, section_counts) -> List[int]:
    """Sorted kept timesteps.  '' / [T] keeps all; 'ddimN' uses the fixed DDIM stride."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            return ddim_timesteps(T, int(section_counts[4:]))
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(T, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        frac = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return sorted(set(steps))


class Schedule:
    """float64 tables of a (possibly respaced) process, exactly as SpacedDiffusion holds them:
    betas are re-derived from the base cumulative products at the kept steps even when
    every step is kept (respace.py:78-85)."""

    def __init__(self, base_betas: np.ndarray, use_timesteps: Optional[Sequence[int]] = None):
        base_betas = np.asarray(base_betas, dtype=np.float64)
        base_abar = np.cumprod(1.0 - base_betas, axis=0)
        keep = set(range(len(base_betas))) if use_timesteps is None else set(use_timesteps)
        last, new_betas, tmap = 1.0, [], []
        for i, ab in enumerate(base_abar):
            if i in keep:
                new_betas.append(1 - ab / last)
                last = ab
                tmap.append(i)
        self.timestep_map = tmap
        b = np.array(new_betas, dtype=np.float64)
        self.betas = b
        self.num_timesteps = len(b)
        a = 1.0 - b
        self.alphas_cumprod = np.cumprod(a, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(a) / (1.0 - self.alphas_cumprod)

    def f32(self, name: str, t: Tensor, like: Tensor) -> Tensor:
        v = torch.from_numpy(getattr(self, name))[t].float()
        while v.dim() < like.dim():
            v = v[..., None]
        return v.expand(like.shape)


def make_schedule(noise_schedule: str = "cosine", T: int = 1000, respacing="") -> Schedule:
    betas = cosine_betas(T) if noise_schedule == "cosine" else linear_betas(T)
    keep = None if not respacing else space_timesteps(T, respacing)
    return Schedule(betas, keep)


ModelFn = Callable[[Tensor, Tensor], Tensor]   # (x, original-scale timesteps) -> x0 prediction


def p_sample(s: Schedule, model: ModelFn, x: Tensor, t: Tensor, z: Tensor, clip_denoised: bool = False) -> Dict[str, Tensor]:
    """One ancestral step, START_X / FIXED_SMALL; z is the injected standard-normal draw."""
    tmap = torch.tensor(s.timestep_map, dtype=t.dtype)
    x0 = model(x, tmap[t])
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = s.f32("posterior_mean_coef1", t, x) * x0 + s.f32("posterior_mean_coef2", t, x) * x
    logvar = s.f32("posterior_log_variance_clipped", t, x)
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return {"sample": mean + nonzero * torch.exp(0.5 * logvar) * z, "pred_xstart": x0}


def ddim_sample(s: Schedule, model: ModelFn, x: Tensor, t: Tensor, z: Tensor, eta: float = 0.0,
                clip_denoised: bool = False) -> Dict[str, Tensor]:
    tmap = torch.tensor(s.timestep_map, dtype=t.dtype)
    x0 = model(x, tmap[t])
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    eps = (s.f32("sqrt_recip_alphas_cumprod", t, x) * x - x0) / s.f32("sqrt_recipm1_alphas_cumprod", t, x)
    ab, abp = s.f32("alphas_cumprod", t, x), s.f32("alphas_cumprod_prev", t, x)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
    nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
    return {"sample": mean + nonzero * sigma * z, "pred_xstart": x0}


def sample_loop(s: Schedule, model: ModelFn, noise: Tensor, sampler: str = "ddpm", eta: float = 0.0,
                clip_denoised: bool = False, record: Optional[Sequence[int]] = None):
    """Runs T' steps with an injected noise stream ``noise[T'+1, B, 1, L]``: row 0 is x_T,
    row 1+k is the draw consumed by loop iteration k (t = T'-1-k).  Returns the final sample
    (and the states listed in ``record`` — loop iteration indices, -1 = x_T)."""
    x = noise[0]
    B = x.shape[0]
    kept = {}
    if record is not None and -1 in record:
        kept[-1] = x.clone()
    step = p_sample if sampler == "ddpm" else (lambda *a, **k: ddim_sample(*a, eta=eta, **k))
    with torch.no_grad():
        for k, i in enumerate(range(s.num_timesteps - 1, -1, -1)):
            t = torch.tensor([i] * B, dtype=torch.long)
            x = step(s, model, x, t, noise[1 + k], clip_denoised=clip_denoised)["sample"]
            if record is not None and k in record:
                kept[k] = x.clone()
    return (x, kept) if record is not None else x

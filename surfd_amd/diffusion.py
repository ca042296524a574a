"""Reverse-diffusion sampler with the reference's call surface.

  get_named_beta_schedule / betas_for_alpha_bar   <- diffusion/gaussian_diffusion.py:23-67
  space_timesteps / SpacedDiffusion / _WrappedModel <- diffusion/respace.py:7-132
  p_mean_variance / p_sample / p_sample_loop(_progressive) / ddim_sample /
  ddim_sample_loop(_progressive) / q_sample        <- diffusion/gaussian_diffusion.py:185-972
  create_gaussian_diffusion                         <- utils/model_util.py:32-67

Only what sampling reaches is implemented (START_X mean, FIXED_SMALL/FIXED_LARGE variance);
training losses, PLMS and classifier guidance (cond_fn) are out of scope.

Two execution modes with identical arithmetic:
  * fused  — the model is the native ``MDM`` (optionally wrapped by ClassifierFreeSampleModel)
             and no per-step Python hook is requested: the whole loop is ONE C call
             (surfd_sample_loop): timestep-only work hoisted for all steps, ~115 kernel
             launches per step enqueued from C++, no Python in the loop;
  * generic — any ``model(x, t, **model_kwargs)`` callable, one Python iteration per step
             (reference behaviour, progress bars, dump_steps, init_image ...).
"""
from __future__ import annotations

import ctypes as C
import math
import time
from copy import deepcopy
from typing import Optional, Sequence

import numpy as np
import torch as th

from . import _native as N


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    out = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        out.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(out)


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.0):
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def _ddim_stride(num_timesteps: int, want: int) -> int:
    """The fixed stride whose range(0, T, stride) has exactly `want` entries ('ddimN' respacing)."""
    hits = [st for st in range(1, num_timesteps) if len(range(0, num_timesteps, st)) == want]
    if not hits:
        raise ValueError(f"cannot create exactly {want} steps with an integer stride out of {num_timesteps}")
    return hits[0]


def space_timesteps(num_timesteps, section_counts):
    """Original timesteps kept by a respacing spec (semantics of diffusion/respace.py:7-60): 'ddimN' keeps every
    stride-th step; a list / comma string splits [0, T) into equal sections (the first T % n one longer) and keeps
    `count` evenly spread steps of each, end points included."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            return set(range(0, num_timesteps, _ddim_stride(num_timesteps, int(section_counts[4:]))))
        section_counts = [int(tok) for tok in section_counts.split(",")]
    n_sec = len(section_counts)
    kept = set()
    first = 0
    for sec, count in enumerate(section_counts):
        length = num_timesteps // n_sec + (1 if sec < num_timesteps % n_sec else 0)
        if count > length:
            raise ValueError(f"cannot divide section of {length} steps into {count}")
        gap = (length - 1) / (count - 1) if count > 1 else 1
        # positions are a RUNNING float sum of the gap, rounded half-to-even — the accumulated rounding is part of
        # the semantics (gap * j lands on the other side of .5 for some j)
        pos = 0.0
        for _ in range(count):
            kept.add(first + round(pos))
            pos += gap
        first += length
    return kept


def _extract(arr: np.ndarray, t: th.Tensor, shape) -> th.Tensor:
    """float64 table -> device -> gather by t -> float32 -> broadcast (gaussian_diffusion.py:1329-1342)."""
    res = th.from_numpy(arr).to(device=t.device)[t].float()
    while res.dim() < len(shape):
        res = res[..., None]
    return res.expand(shape)


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model, self.timestep_map = model, timestep_map
        self.rescale_timesteps, self.original_num_steps = rescale_timesteps, original_num_steps

    def __call__(self, x, ts, **kwargs):
        new_ts = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)


class SpacedDiffusion:
    """Gaussian diffusion over a subset of the base process' timesteps.  ``betas`` is the base
    schedule; the instance tables are re-derived from the base cumulative products at the kept
    steps (also when all steps are kept — respace.py:78-85)."""

    def __init__(self, use_timesteps, *, betas, predict_xstart=True, sigma_small=True,
                 rescale_timesteps=False, clip_value=1.0, **_ignored):
        if not predict_xstart:
            raise NotImplementedError("Surf-D samples with START_X prediction only (model_util.py:35)")
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        self.rescale_timesteps = rescale_timesteps
        self.sigma_small = sigma_small
        self.clip_value = clip_value
        base_ab = np.cumprod(1.0 - np.array(betas, dtype=np.float64), axis=0)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, ab in enumerate(base_ab):
            if i in self.use_timesteps:
                new_betas.append(1 - ab / last)
                last = ab
                self.timestep_map.append(i)
        b = np.array(new_betas, dtype=np.float64)
        assert b.ndim == 1 and (b > 0).all() and (b <= 1).all()
        self.betas = b
        self.num_timesteps = int(b.shape[0])
        alphas = 1.0 - b
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.time_con = []

    # ---- tables ----------------------------------------------------------------------------------
    def _variance_tables(self):
        if self.sigma_small:
            return self.posterior_variance, self.posterior_log_variance_clipped
        v = np.append(self.posterior_variance[1], self.betas[1:])
        return v, np.log(v)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    # ---- single-step maths (generic path) --------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = th.randn_like(x_start)
        return (_extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                + _extract(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        mean = (_extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + _extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, _extract(self.posterior_variance, t, x_t.shape), _extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        if model_kwargs is None:
            model_kwargs = {}
        assert t.shape == (x.shape[0],)
        model_output = self._wrap_model(model)(x, t, **model_kwargs)
        _ = model_kwargs["y"].keys()       # the reference requires model_kwargs['y'] to be a dict (:288)
        var, logvar = self._variance_tables()
        x0 = model_output
        if denoised_fn is not None:
            x0 = denoised_fn(x0)
        if clip_denoised:
            x0 = x0.clamp(-1, 1)
        mean, _, _ = self.q_posterior_mean_variance(x0, x, t)
        return {"mean": mean, "variance": _extract(var, t, x.shape), "log_variance": _extract(logvar, t, x.shape),
                "pred_xstart": x0}

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        return ((_extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t - pred_xstart)
                / _extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape))

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                 const_noise=False, _z=None):
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not on the sampling path")
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = th.randn_like(x) if _z is None else _z
        if const_noise:
            noise = noise[[0]].repeat(x.shape[0], *([1] * (x.dim() - 1)))
        nonzero_mask = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        sample = out["mean"] + nonzero_mask * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0, _z=None):
        if cond_fn is not None:
            raise NotImplementedError("classifier guidance (cond_fn) is not on the sampling path")
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])
        ab = _extract(self.alphas_cumprod, t, x.shape)
        ab_prev = _extract(self.alphas_cumprod_prev, t, x.shape)
        sigma = eta * th.sqrt((1 - ab_prev) / (1 - ab)) * th.sqrt(1 - ab / ab_prev)
        noise = th.randn_like(x) if _z is None else _z
        mean_pred = out["pred_xstart"] * th.sqrt(ab_prev) + th.sqrt(1 - ab_prev - sigma ** 2) * eps
        nonzero_mask = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        return {"sample": mean_pred + nonzero_mask * sigma * noise, "pred_xstart": out["pred_xstart"]}

    # ---- loops ----------------------------------------------------------------------------------------
    def _native_target(self, model):
        """(mdm, guidance) when the loop can run fused, else None."""
        from .mdm import MDM, ClassifierFreeSampleModel
        inner = model.model if isinstance(model, ClassifierFreeSampleModel) else model
        if isinstance(inner, MDM) and next(inner.parameters()).is_cuda:
            return inner
        return None

    @staticmethod
    def _check_cfg_contract(model, model_kwargs):
        """The fused loop evaluates the denoiser once per step (the wrapper's two evaluations are
        bit-identical) but keeps the wrapper's own preconditions."""
        from .mdm import ClassifierFreeSampleModel
        if isinstance(model, ClassifierFreeSampleModel):
            assert model.model.cond_mode in ["text", "action"]
            y = (model_kwargs or {}).get("y", {})
            _ = y["scale"].view(-1, 1, 1)          # KeyError / AttributeError exactly where the reference would fail

    @staticmethod
    def _progress_bar(mdm, T):
        """progress=True for the fused loop (every sample/generate_*.py of the reference passes it, gaussian_diffusion.py:677-681):
        the loop is ONE C call, so the bar is drawn by a thread that polls the device-side loop counter
        (surfd_unet_loop_progress) while the graph replays are in flight.  -> (stop, thread)"""
        import threading
        from tqdm.auto import tqdm
        Lh, h = mdm._native()
        stop = threading.Event()

        def draw():
            bar = tqdm(total=T)
            it = C.c_int(-1)
            seen_reset = False            # the counter of a previous loop (= its T) is still there until this loop's memset runs
            while True:
                done = stop.is_set()
                N.check(Lh.surfd_unet_loop_progress(h, C.byref(it)))
                v = int(it.value)
                if v < T:
                    seen_reset = True
                if done:
                    v = T
                if (seen_reset or done) and v > bar.n:
                    bar.update(min(v, T) - bar.n)
                if done:
                    break
                time.sleep(0.02)
            bar.close()

        th_ = threading.Thread(target=draw, daemon=True)
        th_.start()
        return stop, th_

    def _fused_inputs(self, mdm, shape, sampler, noise, noise_stream, clip_denoised, model_kwargs, eta, device, return_trajectory):
        """Everything surfd_sample_loop(_begin) takes, as device tensors / host tables: -> dict (holds the host tables alive)."""
        B, L = shape[0], shape[-1]
        T = self.num_timesteps
        if noise_stream is None:
            x_T = noise if noise is not None else th.randn(*shape, device=device)
            z = th.randn(T, *shape, device=device)
            noise_stream = th.cat([x_T[None].float(), z], 0)
        noise_stream = noise_stream.to(device=device, dtype=th.float32).contiguous()
        assert noise_stream.shape == (T + 1, *shape), noise_stream.shape
        y = (model_kwargs or {}).get("y", {})
        ctx, cls = mdm.conditioning(y, B)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float64).astype(np.float32)
        _, logvar = self._variance_tables()
        tabs = {k: f32(v) for k, v in dict(coef1=self.posterior_mean_coef1, coef2=self.posterior_mean_coef2, lv=logvar,
                                           sra=self.sqrt_recip_alphas_cumprod, srm1=self.sqrt_recipm1_alphas_cumprod,
                                           ab=self.alphas_cumprod, abp=self.alphas_cumprod_prev).items()}
        fp = lambda a: a.ctypes.data_as(N.c_f32p)
        tmap = np.ascontiguousarray(self.timestep_map, dtype=np.int64)
        if self.rescale_timesteps:
            raise NotImplementedError("rescale_timesteps=True is never used for sampling (model_util.py:42)")
        cfg = N.SamplerCfg(0 if sampler == "ddpm" else 1, T, int(bool(clip_denoised)), float(eta),
                           tmap.ctypes.data_as(N.c_i64p), fp(tabs["coef1"]), fp(tabs["coef2"]), fp(tabs["lv"]),
                           fp(tabs["sra"]), fp(tabs["srm1"]), fp(tabs["ab"]), fp(tabs["abp"]))
        out = th.empty(B, *shape[1:], device=device, dtype=th.float32)
        traj = th.empty(T, B, *shape[1:], device=device, dtype=th.float32) if return_trajectory else None
        return dict(B=B, L=L, T=T, cfg=cfg, tabs=tabs, tmap=tmap, noise=noise_stream, ctx=ctx, cls=cls, out=out, traj=traj)

    def _fused_loop(self, mdm, shape, sampler, noise, noise_stream, clip_denoised, model_kwargs, eta, device,
                    return_trajectory=False, progress=False):
        a = self._fused_inputs(mdm, shape, sampler, noise, noise_stream, clip_denoised, model_kwargs, eta, device, return_trajectory)
        T, out, traj = a["T"], a["out"], a["traj"]
        Lh, h = mdm._native()
        bar = self._progress_bar(mdm, T) if progress else None
        t0 = time.time()
        try:
            N.check(Lh.surfd_sample_loop(h, C.byref(a["cfg"]), N.ptr(a["noise"]), N.ptr(a["ctx"]), N.ptr(a["cls"]), N.ptr(out), N.ptr(traj),
                                         a["B"], a["L"], N.stream()))
            if bar is not None:
                th.cuda.current_stream(device).synchronize()       # the bar ends when the loop has (the reference's loop is synchronous)
        finally:
            if bar is not None:
                bar[0].set(); bar[1].join()
        # one entry per step, like the reference's per-iteration bookkeeping (gaussian_diffusion.py:683, 708): the loop's host-side
        # time shared evenly (its iterations are replays of one graph); without progress=True the call returns when the replays
        # are queued, not when they have run
        dt = (time.time() - t0) / T
        self.time_con.extend([dt] * T)
        return (out, traj) if return_trajectory else out

    def fused_loops_interleaved(self, jobs, sampler="ddpm", clip_denoised=True, eta=0.0, chunk=1, wait_current=True):
        """Several fused reverse loops driven by ONE host thread: every loop is opened (surfd_sample_loop_begin:
        tables, embeddings, the captured iteration), then the thread hands `chunk` iterations at a time to each loop's stream in
        turn until all have been launched, then closes them.  The submission order is a fixed function of (len(jobs), chunk) —
        no host threads racing for the driver's queue lock — and each loop's result is the one surfd_sample_loop gives for
        the same inputs, bit for bit (tests/test_gpu_unet.py).

        jobs: [{"model": MDM (own execution context: `MDM.replica()`), "shape": (B, 1, L), "noise_stream": [T'+1, B, 1, L],
                "stream": torch.cuda.Stream, "model_kwargs": {...} (optional)}, ...]   ->  [latents [B, 1, L] per job]
        Every job's stream waits for the current stream first (wait_current=False: the caller has ordered the streams itself);
        the caller waits on the job streams (or records events) after."""
        chunk = int(chunk)
        if chunk < 1:
            raise ValueError(f"fused_loops_interleaved: chunk must be >= 1 iteration per turn, got {chunk}")
        opened = []
        cur = th.cuda.current_stream()
        T = self.num_timesteps
        t0 = time.time()
        targets = []
        for j in jobs:                                   # nothing is opened before every job has been checked
            self._check_cfg_contract(j["model"], j.get("model_kwargs"))
            mdm = self._native_target(j["model"])
            if mdm is None:
                raise TypeError("fused_loops_interleaved: every job needs an MDM on the GPU")
            if any(mdm is o for o in targets):
                raise ValueError("fused_loops_interleaved: two jobs share one execution context (use MDM.replica())")
            targets.append(mdm)
        outs = []
        ok = False
        try:
            for j, mdm in zip(jobs, targets):
                dev = next(mdm.parameters()).device
                st = j["stream"]
                if wait_current:
                    st.wait_stream(cur)
                with th.cuda.stream(st):
                    a = self._fused_inputs(mdm, tuple(j["shape"]), sampler, j.get("noise"), j.get("noise_stream"), clip_denoised,
                                           j.get("model_kwargs"), eta, dev, False)
                    Lh, h = mdm._native()
                    opened.append((mdm, a, st, Lh, h))   # before begin: a begin that fails half-way has already queued work reading a[...]
                    N.check(Lh.surfd_sample_loop_begin(h, C.byref(a["cfg"]), N.ptr(a["noise"]), N.ptr(a["ctx"]), N.ptr(a["cls"]), None,
                                                       a["B"], a["L"], N.stream()))
            left = [T] * len(opened)
            rem = C.c_int(0)
            while any(left):
                for q, (mdm, a, st, Lh, h) in enumerate(opened):
                    if left[q]:
                        with th.cuda.stream(st):
                            N.check(Lh.surfd_sample_loop_run(h, chunk, C.byref(rem), N.stream()))
                        if int(rem.value) >= left[q]:
                            raise RuntimeError("fused_loops_interleaved: surfd_sample_loop_run made no progress")
                        left[q] = int(rem.value)
            for mdm, a, st, Lh, h in opened:
                with th.cuda.stream(st):
                    N.check(Lh.surfd_sample_loop_end(h, N.ptr(a["out"]), N.stream()))
                outs.append(a["out"])
            ok = True
        finally:
            # the replays queued so far read noise / ctx / cls on the job streams whether or not every call succeeded: the caching
            # allocator must not hand that memory out again before those streams have passed this point
            for mdm, a, st, Lh, h in opened:
                for k in ("noise", "ctx", "cls"):
                    if a[k] is not None:
                        a[k].record_stream(st)
            if not ok:
                # a loop that was opened and not closed: let what is in flight drain, then abandon it (the next begin on the
                # handle drops the open loop's state, csrc/sampler.hip)
                for mdm, a, st, Lh, h in opened:
                    try:
                        st.synchronize()
                    except Exception:
                        pass
        self.time_con.extend([(time.time() - t0) / T] * T)
        return outs

    def _loop(self, sampler, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
              skip_timesteps, init_image, randomize_class, eta, const_noise, noise_stream):
        """Generic per-step generator behind both *_progressive loops (behaviour of gaussian_diffusion.py:635-708 /
        :908-972): yields the step dict of every iteration, t running from T'-1-skip down to 0."""
        device = device if device is not None else next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        B = shape[0]
        # start state: injected noise row, caller's noise, or a fresh draw
        x = noise_stream[0].to(device) if noise_stream is not None else (noise if noise is not None else th.randn(*shape, device=device))
        steps = range(self.num_timesteps - skip_timesteps - 1, -1, -1)
        if skip_timesteps and init_image is None:
            init_image = th.zeros_like(x)
        if init_image is not None:
            # partial chain: start from q(x_t | init_image) at the first timestep actually run
            x = self.q_sample(init_image, th.full((B,), steps[0], device=device, dtype=th.long), x)
        if progress:
            from tqdm.auto import tqdm
            steps = tqdm(steps)
        step_fn = self.p_sample if sampler == "ddpm" else self.ddim_sample
        for k, i in enumerate(steps):
            began = time.time()
            if randomize_class and "y" in model_kwargs:
                yk = model_kwargs["y"]
                model_kwargs["y"] = th.randint(low=0, high=model.num_classes, size=yk.shape, device=yk.device)
            extra = {"const_noise": const_noise} if sampler == "ddpm" else {"eta": eta}
            with th.no_grad():
                out = step_fn(model, x, th.full((B,), i, device=device, dtype=th.long), clip_denoised=clip_denoised,
                              denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                              _z=None if noise_stream is None else noise_stream[1 + k].to(device), **extra)
                yield out
                x = out["sample"]
            self.time_con.append(time.time() - began)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False, const_noise=False, noise_stream=None):
        if cond_fn_with_grad:
            raise NotImplementedError("cond_fn_with_grad is not on the sampling path")
        yield from self._loop("ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
                              skip_timesteps, init_image, randomize_class, 0.0, const_noise, noise_stream)

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0,
                                     init_image=None, randomize_class=False, cond_fn_with_grad=False, noise_stream=None):
        if cond_fn_with_grad:
            raise NotImplementedError("cond_fn_with_grad is not on the sampling path")
        yield from self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device, progress,
                              skip_timesteps, init_image, randomize_class, eta, False, noise_stream)

    def _can_fuse(self, model, denoised_fn, cond_fn, skip_timesteps, init_image, randomize_class, dump_steps, const_noise,
                  progress, fused):
        if fused is False:
            return None
        plain = (denoised_fn is None and cond_fn is None and not skip_timesteps and init_image is None
                 and not randomize_class and dump_steps is None and not const_noise)
        mdm = self._native_target(model) if plain else None
        if fused is True and mdm is None:
            raise RuntimeError("fused=True needs the native MDM on a GPU and no per-step Python hooks")
        return mdm

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False, skip_timesteps=0, init_image=None, randomize_class=False,
                      cond_fn_with_grad=False, dump_steps=None, const_noise=False, noise_stream=None, fused=None,
                      return_trajectory=False):
        """Same signature as the reference (gaussian_diffusion.py:570-633) plus extensions:
        ``noise_stream`` [T+1,B,1,L] injects every random draw (row 0 = x_T) for reproducible
        parity runs; ``fused`` forces (True) or forbids (False) the single-C-call loop;
        ``return_trajectory`` (fused only) also returns x after every iteration [T,B,1,L]."""
        mdm = self._can_fuse(model, denoised_fn, cond_fn, skip_timesteps, init_image, randomize_class, dump_steps,
                             const_noise, progress, fused)
        if mdm is not None:
            self._check_cfg_contract(model, model_kwargs)
            dev = device if device is not None else next(model.parameters()).device
            return self._fused_loop(mdm, tuple(shape), "ddpm", noise, noise_stream, clip_denoised, model_kwargs, 0.0, dev,
                                    return_trajectory, progress=progress)
        final, dump = None, []
        for i, sample in enumerate(self.p_sample_loop_progressive(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, device=device, progress=progress, skip_timesteps=skip_timesteps,
                init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad,
                const_noise=const_noise, noise_stream=noise_stream)):
            if dump_steps is not None and i in dump_steps:
                dump.append(deepcopy(sample["sample"]))
            final = sample
        return dump if dump_steps is not None else final["sample"]

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                         noise_stream=None, fused=None):
        if dump_steps is not None or const_noise:
            raise NotImplementedError()
        mdm = self._can_fuse(model, denoised_fn, cond_fn, skip_timesteps, init_image, randomize_class, None, False,
                             progress, fused)
        if mdm is not None:
            self._check_cfg_contract(model, model_kwargs)
            dev = device if device is not None else next(model.parameters()).device
            return self._fused_loop(mdm, tuple(shape), "ddim", noise, noise_stream, clip_denoised, model_kwargs, eta, dev, progress=progress)
        final = None
        for sample in self.ddim_sample_loop_progressive(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, device=device, progress=progress, eta=eta, skip_timesteps=skip_timesteps,
                init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad,
                noise_stream=noise_stream):
            final = sample
        return final["sample"]


def create_gaussian_diffusion(args, timestep_respacing=""):
    """utils/model_util.py:32-67 with the respacing knob exposed (the reference hard-codes '')."""
    steps = 1000
    betas = get_named_beta_schedule(args.noise_schedule, steps, 1.0)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas, predict_xstart=True,
                           sigma_small=getattr(args, "sigma_small", True), rescale_timesteps=False,
                           clip_value=getattr(args, "clip_value", 1.0))

#!/usr/bin/env python3
"""Golden-vector generator: runs ONLY in the build container, where the reference is
mounted at /root/reference.  It imports the reference's own Python modules (never copies
them), feeds them the deterministic synthetic weights of surfd_amd.synth, and writes small
input/output fixtures to tests/golden/*.npz.  The fixtures are data only.

    python tools/make_golden.py [--only g3,g6] [--ref /root/reference]

Import recipe: stub the packages that are not installed and that the hot
path never calls (open3d, clip, trimesh, pymeshlab, the Cython marching-cubes module) and
neutralise the hard-coded ``.cuda()`` calls of meshudf.py so everything runs on CPU.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")

from surfd_amd.spec import DecoderConfig, UNetConfig  # noqa: E402
from surfd_amd import synth  # noqa: E402


def import_reference(ref: str):
    sys.path.insert(0, ref)
    for name in ["open3d", "clip", "trimesh", "pymeshlab", "meshudf._marching_cubes_lewiner"]:
        sys.modules[name] = MagicMock()
    torch.Tensor.cuda = lambda self, *a, **k: self          # meshudf.py:78,99,113,121,215,236
    mods = types.SimpleNamespace()
    from utils import model_util                              # noqa
    from diffusion import gaussian_diffusion, respace         # noqa
    from models import openaimodel, cfg_sampler               # noqa
    from utils import ldm_utils                               # noqa
    from AutoEncoder.models import cbndec, coordsenc          # noqa
    from meshudf import meshudf                               # noqa
    mods.model_util, mods.gd, mods.respace = model_util, gaussian_diffusion, respace
    mods.openaimodel, mods.ldm_utils, mods.cfg_sampler = openaimodel, ldm_utils, cfg_sampler
    mods.cbndec, mods.coordsenc, mods.meshudf = cbndec, coordsenc, meshudf
    return mods


def ref_args(cond_mode: str):
    return types.SimpleNamespace(cond_mode=cond_mode, arch="OpenUNet", num_actions=9, dataset="deepfashion3d",
                                 noise_schedule="cosine", sigma_small=True, clip_value=1.0)


class FakeClip(torch.nn.Module):
    """Stands in for the CLIP text tower (weights unavailable offline; the tower is an input producer outside the
    path): ``encode_text`` returns the rows of a fixed table selected by the 'token' ids ``tokenize`` hands out."""

    def __init__(self, table):
        super().__init__()
        self.register_buffer("table", table)

    def encode_text(self, tokens):
        return self.table[tokens]


def install_fake_clip(table):
    clip = sys.modules["clip"]
    clip.load = lambda *a, **k: (FakeClip(table), None)
    clip.tokenize = lambda raw_text, truncate=True: torch.tensor([int(t) for t in raw_text], dtype=torch.long)


def build_model(R, cond_mode: str, head_gain: float = 1.0):
    model, diffusion = R.model_util.create_model_and_diffusion(ref_args(cond_mode))
    cfg = UNetConfig(num_classes=9 if "category" in cond_mode else None)
    sd = synth.synth_unet_state_dict(cfg, head_gain=head_gain)
    ref_sd = {k: v for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    assert list(ref_sd.keys()) == list(sd.keys()), "spec.py key order differs from the reference"
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
    R.model_util.load_model_wo_clip(model, sd)
    model.eval()            # MDM.train() returns None (models/mdm.py:112-113)
    return model, diffusion


def build_decoder(R, D: int):
    dec = R.cbndec.CbnDecoder(63, D, 512, 5)
    sd = synth.synth_decoder_state_dict(DecoderConfig(latent_dim=D))
    assert list(dec.state_dict().keys()) == list(sd.keys())
    dec.load_state_dict(sd, strict=True)
    return dec.eval()


def save(name: str, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KB)")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------
def g1_g2(R):
    t = torch.tensor([0, 1, 500, 999])
    save("g1_timestep_embedding", t=t, emb=R.ldm_utils.timestep_embedding(t, 224))
    d = R.model_util.create_gaussian_diffusion(ref_args("no_cond"))
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    arrs = {n: getattr(d, n) for n in names}
    base = R.gd.get_named_beta_schedule("cosine", 1000, 1.0)
    dd = R.respace.SpacedDiffusion(use_timesteps=R.respace.space_timesteps(1000, "ddim50"), betas=base,
                                   model_mean_type=R.gd.ModelMeanType.START_X,
                                   model_var_type=R.gd.ModelVarType.FIXED_SMALL, loss_type=R.gd.LossType.MSE,
                                   rescale_timesteps=False, args=ref_args("no_cond"))
    for n in names:
        arrs["ddim50_" + n] = getattr(dd, n)
    arrs["ddim50_timestep_map"] = np.array(dd.timestep_map)
    arrs["full_timestep_map"] = np.array(d.timestep_map)
    arrs["base_betas"] = base
    arrs["sections_10_15_20_of_300"] = np.array(sorted(R.respace.space_timesteps(300, [10, 15, 20])))
    save("g2_schedule", **arrs)


def g3_g4(R):
    # full forwards + per-module activations captured with forward hooks
    model, _ = build_model(R, "no_cond")
    x = rnd((2, 1, 32), 11)
    t = torch.tensor([999, 10])
    captured = {}

    def hook(name):
        def fn(mod, inp, out):
            captured[name + ".in"] = inp[0].detach().clone()
            if len(inp) > 1 and isinstance(inp[1], torch.Tensor):
                captured[name + ".emb"] = inp[1].detach().clone()
            captured[name + ".out"] = out.detach().clone()
        return fn

    U = model.Unet
    targets = {
        "input_blocks.1.0": U.input_blocks[1][0], "input_blocks.1.1": U.input_blocks[1][1],
        "input_blocks.3.0": U.input_blocks[3][0], "input_blocks.4.0": U.input_blocks[4][0],
        "middle_block.1": U.middle_block[1], "output_blocks.0.0": U.output_blocks[0][0],
        "output_blocks.2.1": U.output_blocks[2][1], "output_blocks.5.0": U.output_blocks[5][0],
        "out": U.out,
    }
    hs = [m.register_forward_hook(hook(n)) for n, m in targets.items()]
    # checkpoint() re-enters modules under no_grad only; hooks on the module still fire once
    with torch.no_grad():
        y = model(x, t, y={})
    for h in hs:
        h.remove()
    save("g3_unet_nocond_L32", x=x, t=t, out=y)
    save("g4_modules_nocond_L32", **{k.replace(".", "__"): v for k, v in captured.items()})

    model, _ = build_model(R, "img")
    x = rnd((2, 1, 64), 12)
    ctx = rnd((2, 512), 13, 0.3)
    with torch.no_grad():
        y = model(x, torch.tensor([777, 3]), y={"context": ctx})
    save("g3_unet_ctx_L64", x=x, t=torch.tensor([777, 3]), context=ctx, out=y)

    model, _ = build_model(R, "category")
    x = rnd((2, 1, 32), 14)
    lab = torch.tensor([3, 7])
    with torch.no_grad():
        y = model(x, torch.tensor([500, 0]), y={"action_text": lab})
    save("g3_unet_category_L32", x=x, t=torch.tensor([500, 0]), labels=lab, out=y)


class NoiseFeeder:
    """Replaces th.randn_like inside the reference's gaussian_diffusion module."""
    def __init__(self, stream):
        self.stream, self.k = stream, 0

    def __call__(self, x):
        z = self.stream[self.k]
        self.k += 1
        assert z.shape == x.shape
        return z.clone()


def g5_g6(R):
    model, diff = build_model(R, "no_cond")
    th = R.gd.th
    real = th.randn_like
    try:
        # G5: single steps with injected z
        x = rnd((2, 1, 32), 21)
        out = {"x": x}
        for tt in [999, 500, 1, 0]:
            z = rnd((2, 1, 32), 100 + tt)
            th.randn_like = NoiseFeeder([z])
            with torch.no_grad():
                r = diff.p_sample(model, x, torch.tensor([tt, tt]), clip_denoised=False, model_kwargs={"y": {}})
            out[f"z_{tt}"], out[f"sample_{tt}"], out[f"x0_{tt}"] = z, r["sample"], r["pred_xstart"]
        base = R.gd.get_named_beta_schedule("cosine", 1000, 1.0)
        dd = R.respace.SpacedDiffusion(use_timesteps=R.respace.space_timesteps(1000, "ddim50"), betas=base,
                                       model_mean_type=R.gd.ModelMeanType.START_X,
                                       model_var_type=R.gd.ModelVarType.FIXED_SMALL, loss_type=R.gd.LossType.MSE,
                                       rescale_timesteps=False, args=ref_args("no_cond"))
        for tt, eta in [(49, 0.0), (25, 0.0), (0, 0.0), (25, 0.7)]:
            z = rnd((2, 1, 32), 200 + tt)
            th.randn_like = NoiseFeeder([z])
            with torch.no_grad():
                r = dd.ddim_sample(model, x, torch.tensor([tt, tt]), clip_denoised=False, model_kwargs={"y": {}}, eta=eta)
            tag = f"{tt}_eta{int(eta * 10)}"
            out[f"ddim_z_{tag}"], out[f"ddim_sample_{tag}"] = z, r["sample"]
        save("g5_single_steps", **out)

        # G6a: 50-step DDIM trajectory (config C1: B=1, L=32, eta=0)
        noise = synth.synth_noise_batch(50, 0, 1, 32)
        th.randn_like = NoiseFeeder(list(noise[1:]))
        rec = {}
        with torch.no_grad():
            for k, o in enumerate(dd.ddim_sample_loop_progressive(model, (1, 1, 32), noise=noise[0].clone(),
                                                                  clip_denoised=False, model_kwargs={"y": {}}, eta=0.0)):
                if k in (0, 24, 48, 49):
                    rec[f"x_after_{k}"] = o["sample"].clone()
        save("g6_ddim50_B1_L32", seed=1234, **rec)

        # G6b: 1000-step DDPM trajectory (B=2, L=32)
        noise = synth.synth_noise_batch(1000, 0, 2, 32)
        th.randn_like = NoiseFeeder(list(noise[1:]))
        rec = {}
        t0 = time.time()
        with torch.no_grad():
            for k, o in enumerate(diff.p_sample_loop_progressive(model, (2, 1, 32), noise=noise[0].clone(),
                                                                 clip_denoised=False, model_kwargs={"y": {}})):
                if k in (0, 1, 499, 998, 999):
                    rec[f"x_after_{k}"] = o["sample"].clone()
        print(f"  reference 1000-step DDPM loop B=2: {time.time() - t0:.1f}s")
        save("g6_ddpm1000_B2_L32", seed=1234, **rec)
    finally:
        th.randn_like = real


def make_ref_udf(R, dec, lat):
    enc = R.coordsenc.CoordsEncoder()

    def udf_func(c):
        e = enc.encode(c.unsqueeze(0))
        p = dec(e, lat).squeeze(0)
        p = torch.sigmoid(p)
        return (1 - p) * 0.1
    return udf_func


def g7_g8(R):
    enc = R.coordsenc.CoordsEncoder()
    pts = torch.rand(16, 3, generator=torch.Generator().manual_seed(31)) * 2 - 1
    save("g7_encode", pts=pts, enc=enc.encode(pts))
    for D in (32, 64):
        dec = build_decoder(R, D)
        lat = rnd((1, D), 40 + D, 0.8)
        pts = torch.rand(4096, 3, generator=torch.Generator().manual_seed(41 + D)) * 2 - 1
        f = make_ref_udf(R, dec, lat)
        with torch.no_grad():
            logit = dec(enc.encode(pts.unsqueeze(0)), lat).squeeze(0)
        udf = R.meshudf.sample_udf(f, pts, 2 ** 16)
        grads = R.meshudf.sample_grads(f, pts, 2 ** 12)
        save(f"g8_decoder_D{D}", lat=lat, pts=pts, logit=logit, udf=udf, ngrad=grads)


def g8_direction_flips(R):
    """The number behind the gradient-direction tolerance (tests/test_gpu_decoder_grid.py:_check_directions): how far the
    REFERENCE's own fp32 sample_grads is from an fp64 evaluation of the same decoder on the G8 points.  The field is piecewise
    linear in 11 x 512 ReLU units; a point whose pre-activation sits within fp32 rounding of a kink takes the other branch in a
    different — equally valid — evaluation, and its direction jumps.  Written as data (fractions, worst cosine) per decoder."""
    import copy
    import json
    out = {}
    for D in (32, 64):
        dec = build_decoder(R, D)
        for prm in dec.parameters():
            prm.requires_grad = False
        lat = rnd((1, D), 40 + D, 0.8)
        pts = torch.rand(4096, 3, generator=torch.Generator().manual_seed(41 + D)) * 2 - 1
        g32 = R.meshudf.sample_grads(make_ref_udf(R, dec, lat), pts, 2 ** 12)
        dec64 = copy.deepcopy(dec).double()
        g64 = R.meshudf.sample_grads(make_ref_udf(R, dec64, lat.double()), pts.double(), 2 ** 12)
        cos = (g32.double() * g64.double()).sum(-1)
        ok = g64.norm(dim=-1) > 0.5
        cos = cos[ok]
        out[f"D{D}"] = {"points": int(ok.sum()), "frac_cos_le_1m1e-5": float((cos <= 1 - 1e-5).double().mean()),
                        "frac_cos_le_1m1e-3": float((cos <= 1 - 1e-3).double().mean()), "worst_cosine": float(cos.min()),
                        "median_one_minus_cos": float((1 - cos).median())}
        print(f"  D={D}: reference fp32 autograd vs fp64 on {int(ok.sum())} points: cos <= 1-1e-5 on "
              f"{100 * out[f'D{D}']['frac_cos_le_1m1e-5']:.3f} %, worst cosine {out[f'D{D}']['worst_cosine']:.6f}")
    out["what"] = ("reference meshudf.sample_grads in fp32 (as the reference runs it) against the same call on a .double() copy of the decoder, "
                   "G8 points (4096 uniform in [-1,1]^3), synthetic decoder weights; cosine between the two -normalize(grad) vectors")
    path = os.path.join(OUT, "g8_direction_flips.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"  wrote {path}")


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def g9(R):
    dec = build_decoder(R, 32)
    lat = rnd((1, 32), 51, 0.8)
    f = make_ref_udf(R, dec, lat)
    calls = []

    def counting(c):
        calls.append(c.shape[0])
        return f(c)
    t0 = time.time()
    gf = R.meshudf.GridFiller(64)
    udf, grads = gf.fill_grid(counting, 2 ** 12)
    print(f"  reference GridFiller(64) with synthetic decoder: {time.time() - t0:.1f}s, {sum(calls)} queries")
    udf, grads = udf.detach(), grads.detach()
    sub = torch.arange(0, 64 ** 3, 5)
    save("g9_grid64_decoder", lat=lat, udf=udf, grad_idx=sub, grad_sub=grads.reshape(-1, 3)[sub],
         grad_nonzero=int((grads.abs().sum(-1) > 0).sum()), udf_sum=float(udf.double().sum()))


def g10(R, sizes):
    from oracle.gridfiller import analytic_field
    res = {}
    for N in sizes:
        per_call = []

        def counting(c):
            per_call.append(c.shape[0])
            return analytic_field(c)
        t0 = time.time()
        gf = R.meshudf.GridFiller(N)
        udf, grads = gf.fill_grid(counting, 2 ** 30)      # one call per level, then one gradient call
        nl = len(gf.N_levels)
        res[f"N{N}_fwd_per_level"] = np.array(per_call[:nl])
        res[f"N{N}_grad_points"] = np.array(sum(per_call[nl:]))
        res[f"N{N}_udf_sum"] = np.array(float(udf.double().sum()))
        res[f"N{N}_udf_sha256"] = np.array(sha(udf))
        res[f"N{N}_grad_abs_sum"] = np.array(float(grads.double().abs().sum()))
        res[f"N{N}_grad_nonzero"] = np.array(int((grads != 0).any(-1).sum()))
        if N <= 64:
            res[f"N{N}_udf"] = udf.numpy()
            res[f"N{N}_grads_f16"] = grads.numpy().astype(np.float16)
        print(f"  reference GridFiller({N}) analytic: {time.time() - t0:.1f}s  fwd={per_call[:nl]} grad={sum(per_call[nl:])}")
        del gf, udf, grads
    save("g10_grid_analytic", **res)



def run_loop(R, model, diff, shape, noise, sampler, model_kwargs, record, eta=0.0):
    """Reference loop with the injected noise stream; returns {x_after_k} for k in record."""
    th = R.gd.th
    real = th.randn_like
    th.randn_like = NoiseFeeder(list(noise[1:]))
    rec = {}
    try:
        with torch.no_grad():
            if sampler == "ddim":
                it = diff.ddim_sample_loop_progressive(model, shape, noise=noise[0].clone(), clip_denoised=False,
                                                       model_kwargs=model_kwargs, eta=eta)
            else:
                it = diff.p_sample_loop_progressive(model, shape, noise=noise[0].clone(), clip_denoised=False,
                                                    model_kwargs=model_kwargs)
            for k, o in enumerate(it):
                if k in record:
                    rec[f"x_after_{k}"] = o["sample"].clone()
    finally:
        th.randn_like = real
    return rec


def ddim50(R):
    base = R.gd.get_named_beta_schedule("cosine", 1000, 1.0)
    return R.respace.SpacedDiffusion(use_timesteps=R.respace.space_timesteps(1000, "ddim50"), betas=base,
                                     model_mean_type=R.gd.ModelMeanType.START_X,
                                     model_var_type=R.gd.ModelVarType.FIXED_SMALL, loss_type=R.gd.LossType.MSE,
                                     rescale_timesteps=False, args=ref_args("no_cond"))


def g11_conditioned_loops(R):
    """Configs C4 / C5: L=64, context[B,512]; C5 = 'img' mode, C4 = 'text' mode under the
    classifier-free wrapper with scale 3.0 (the CLIP tower replaced by a fixed embedding table)."""
    B, L = 8, 64
    ctx = synth.synth_context(0, B)
    gain = synth.CONTRACTIVE_HEAD_GAIN      # contractive head: the 50-step trajectory is comparable to ~1e-6, not 1e-3
    # C5: image-conditioned, 50-step DDIM (eta 0)
    model, _ = build_model(R, "img", head_gain=gain)
    noise = synth.synth_noise_batch(50, 0, B, L)
    t0 = time.time()
    rec = run_loop(R, model, ddim50(R), (B, 1, L), noise, "ddim", {"y": {"context": ctx}}, (0, 24, 49))
    print(f"  reference img-cond DDIM-50 B={B} L={L}: {time.time() - t0:.1f}s")
    save("g11_ddim50_img_B8_L64", seed=1234, ctx_seed=77, head_gain=gain, **rec)
    # C4: text-conditioned + CFG wrapper, 50-step DDIM
    install_fake_clip(ctx)
    model, _ = build_model(R, "text", head_gain=gain)
    wrapped = R.cfg_sampler.ClassifierFreeSampleModel(model)
    kw = {"y": {"text": [str(i) for i in range(B)], "scale": torch.ones(B) * 3.0}}
    t0 = time.time()
    rec = run_loop(R, wrapped, ddim50(R), (B, 1, L), noise, "ddim", kw, (0, 24, 49))
    print(f"  reference text+CFG DDIM-50 B={B} L={L}: {time.time() - t0:.1f}s")
    save("g11_ddim50_textcfg_B8_L64", seed=1234, ctx_seed=77, scale=3.0, head_gain=gain, **rec)


def g12_contractive(R):
    """1000-step DDPM chains that CAN be compared end to end: the head of the synthetic denoiser is scaled by
    synth.CONTRACTIVE_HEAD_GAIN, which makes the map x_t -> x_{t-1} contractive (a 1e-4 perturbation of x_T
    shrinks to 1e-7 at the end instead of growing to O(1))."""
    g = synth.CONTRACTIVE_HEAD_GAIN
    keep = (0, 1, 499, 998, 999)
    model, diff = build_model(R, "no_cond", head_gain=g)
    noise = synth.synth_noise_batch(1000, 0, 2, 32)
    t0 = time.time()
    rec = run_loop(R, model, diff, (2, 1, 32), noise, "ddpm", {"y": {}}, keep)
    print(f"  reference contractive 1000-step DDPM B=2 L=32: {time.time() - t0:.1f}s")
    save("g12_ddpm1000_contractive_B2_L32", seed=1234, head_gain=g, **rec)
    # C4 at full length: text + CFG(3.0), L=64, 1000 steps
    B, L = 2, 64
    ctx = synth.synth_context(0, B)
    install_fake_clip(ctx)
    model, diff = build_model(R, "text", head_gain=g)
    wrapped = R.cfg_sampler.ClassifierFreeSampleModel(model)
    kw = {"y": {"text": [str(i) for i in range(B)], "scale": torch.ones(B) * 3.0}}
    noise = synth.synth_noise_batch(1000, 0, B, L)
    t0 = time.time()
    rec = run_loop(R, wrapped, diff, (B, 1, L), noise, "ddpm", kw, keep)
    print(f"  reference contractive text+CFG 1000-step DDPM B=2 L=64: {time.time() - t0:.1f}s")
    save("g12_ddpm1000_contractive_textcfg_B2_L64", seed=1234, ctx_seed=77, scale=3.0, head_gain=g, **rec)


def g9_d64(R):
    dec = build_decoder(R, 64)
    lat = rnd((1, 64), 52, 0.8)
    f = make_ref_udf(R, dec, lat)
    t0 = time.time()
    gf = R.meshudf.GridFiller(64)
    udf, grads = gf.fill_grid(f, 2 ** 12)
    print(f"  reference GridFiller(64) with the D=64 synthetic decoder: {time.time() - t0:.1f}s")
    udf, grads = udf.detach(), grads.detach()
    sub = torch.arange(0, 64 ** 3, 5)
    save("g9_grid64_decoder_D64", lat=lat, udf=udf, grad_idx=sub, grad_sub=grads.reshape(-1, 3)[sub],
         grad_nonzero=int((grads.abs().sum(-1) > 0).sum()), udf_sum=float(udf.double().sum()))



def g13_marching_cubes(R, with_512: bool):
    """G11 of SURVEY.md §8c: the reference's own compiled UDF marching cubes (oracle/build_ref.py builds
    meshudf/_marching_cubes_lewiner_cy.pyx into oracle/_ref/) on deterministic synthetic grids (tests/mc_fields.py):
    counts + SHA-256 of the vertex / face arrays for every case, full arrays for the small ones."""
    import base64
    import importlib.util
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import mc_fields
    from oracle import build_ref
    cy = build_ref.load()
    assert cy is not None, "oracle/_ref could not be built"
    spec = importlib.util.spec_from_file_location("ref_luts", os.path.join("/root/reference", "meshudf", "_marching_cubes_lewiner_luts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tables = {n: np.frombuffer(base64.decodebytes(getattr(mod, n)[1].encode()), dtype=np.int8).reshape(getattr(mod, n)[0])
              for n in build_ref.LUT_ORDER[3:]}
    tables["EDGESRELX"] = np.array([[0, 1], [1, 1], [1, 0], [0, 0], [0, 1], [1, 1], [1, 0], [0, 0], [0, 0], [1, 1], [1, 1], [0, 0]], np.int8)
    tables["EDGESRELY"] = np.array([[0, 0], [0, 1], [1, 1], [1, 0], [0, 0], [0, 1], [1, 1], [1, 0], [0, 0], [0, 0], [1, 1], [1, 1]], np.int8)
    tables["EDGESRELZ"] = np.array([[0, 0], [0, 0], [0, 0], [0, 0], [1, 1], [1, 1], [1, 1], [1, 1], [0, 1], [0, 1], [0, 1], [0, 1]], np.int8)
    cases = [("two_spheres", 48), ("two_spheres", 96), ("open_sheet", 48), ("open_sheet", 96), ("noisy_blob", 48),
             ("noisy_blob", 96), ("thin_shell", 64), ("thin_shell", 128), ("thin_shell", 256)]
    if with_512:
        cases.append(("thin_shell", 512))
    out = {}
    for name, N in cases:
        udf, grads = mc_fields.FIELDS[name](N)
        t0 = time.time()
        v, f, n, val = build_ref.reference_udf_mc(cy, tables, udf, grads)
        tag = f"{name}_{N}"
        out[tag + "_nv"], out[tag + "_nf"] = np.array(len(v)), np.array(len(f))
        # the INPUT grid's hash: sin / sqrt of the math libraries differ in the last bit between CPUs, so a test on
        # another host first checks that it is meshing the very same grid before comparing output hashes
        out[tag + "_input_sha256"] = np.array(hashlib.sha256(udf.tobytes() + grads.tobytes()).hexdigest())
        out[tag + "_verts_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).hexdigest())
        out[tag + "_faces_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(f, np.int32).tobytes()).hexdigest())
        if N <= 48:
            out[tag + "_verts"], out[tag + "_faces"] = np.ascontiguousarray(v, np.float32), np.ascontiguousarray(f, np.int32)
            out[tag + "_normals"] = np.ascontiguousarray(n, np.float32)
        print(f"  reference marching_cubes_udf {tag}: {len(v)} verts / {len(f)} faces in {time.time() - t0:.2f}s")
    save("g13_marching_cubes", **out)


def g13_lut_hashes(R):
    """SHA-256 + shape of every case table of the reference's meshudf/_marching_cubes_lewiner_luts.py (base64 int8
    blobs) and of the three edge-corner tables its driver builds (_marching_cubes_lewiner.py:171-224): pins the
    numbers compiled into surfd_amd/csrc/mc_luts.h on ANY host, also where the reference tree is absent."""
    import base64
    import importlib.util
    from oracle import build_ref
    spec = importlib.util.spec_from_file_location("ref_luts", os.path.join("/root/reference", "meshudf", "_marching_cubes_lewiner_luts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for n in build_ref.LUT_ORDER[3:]:
        shape, text = getattr(mod, n)
        ar = np.frombuffer(base64.decodebytes(text.encode()), dtype=np.int8).reshape(shape)
        out[n + "_sha256"] = np.array(hashlib.sha256(np.ascontiguousarray(ar).tobytes()).hexdigest())
        out[n + "_shape"] = np.array(ar.shape)
    # the driver's edge -> corner-offset tables: the three module-level assignments, evaluated from the reference source
    import ast
    tree = ast.parse(open(os.path.join("/root/reference", "meshudf", "_marching_cubes_lewiner.py")).read())
    ns = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "").startswith("EDGETORELATIVEPOS"):
            ns[node.targets[0].id] = np.array(ast.literal_eval(node.value.args[0]), np.int8)
    for ours, theirs in (("EDGESRELX", "EDGETORELATIVEPOSX"), ("EDGESRELY", "EDGETORELATIVEPOSY"), ("EDGESRELZ", "EDGETORELATIVEPOSZ")):
        ar = np.ascontiguousarray(np.asarray(ns[theirs]), np.int8)
        out[ours + "_sha256"] = np.array(hashlib.sha256(ar.tobytes()).hexdigest())
        out[ours + "_shape"] = np.array(ar.shape)
    save("g13_lut_sha256", **out)


def g15_image_preprocess(R):
    """f4: sample/generate_image.py:92-107's mask / crop preprocessing with the reference's OWN functions
    (data_loaders/dataset.py:19-77; the module imports torchvision at the top, absent here, so the two function
    definitions are evaluated from the reference source with numpy and PIL, which is all they use) on seeded images
    whose masks touch every border case of the padding logic."""
    import ast
    from PIL import Image
    src = open(os.path.join("/root/reference", "data_loaders", "dataset.py")).read()
    tree = ast.parse(src)
    ns = {"np": np, "Image": Image}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("mask2bbox", "crop_square"):
            exec(compile(ast.Module([node], []), "dataset.py", "exec"), ns)
    rng = np.random.default_rng(15)
    cases = {"center": (300, 400, (120, 90, 260, 210)), "tall_left": (240, 320, (0, 10, 60, 230)), "wide_bottom": (200, 360, (40, 150, 359, 199)),
             "corner": (180, 180, (120, 0, 179, 70)), "whole": (128, 160, (0, 0, 159, 127)), "thin": (256, 256, (100, 30, 103, 220))}
    out = {}
    for name, (h, w, (x0, y0, x1, y1)) in cases.items():
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        # smooth it a little so that the bicubic resize is not pure noise
        img = (0.5 * img + 0.5 * np.roll(img, 3, axis=1)).astype(np.uint8)
        mask = np.zeros((h, w), dtype=bool)
        yy, xx = np.mgrid[0:h, 0:w]
        mask[(yy >= y0) & (yy <= y1) & (xx >= x0) & (xx <= x1) & (((yy + xx) % 7) != 0)] = True
        mask[y0, x0] = mask[y1, x1] = True
        bbox = ns["mask2bbox"](mask)
        r = 0.7
        comp = img * mask[:, :, None] + (1 - mask[:, :, None]) * (r * 255 + (1 - r) * img)
        clean = img * mask[:, :, None]
        out[name + "__img"], out[name + "__mask"] = img, mask
        out[name + "__bbox"] = np.array([int(v) for v in bbox])
        for tag, arr in (("comp", comp), ("clean", clean)):
            res = np.ascontiguousarray(np.array(ns["crop_square"](arr.astype(np.uint8), list(bbox))))
            out[f"{name}__{tag}_sha256"] = np.array(hashlib.sha256(res.tobytes()).hexdigest())      # the whole 256 x 256 x 3 crop
            out[f"{name}__{tag}_sub"] = res[::8, ::8].copy()                                          # every 8th pixel, for a readable failure
    save("g15_image_preprocess", **out)


def g16_clip_towers(R):
    """f4: the reference's vendored CLIP model class (CLIP/clip/model.py) as ViT-B/32 with the seeded weights of
    surfd_amd.synth.synth_clip_state_dict: encode_image on one seeded 224 x 224 picture batch, encode_text on token
    rows.  Tokens: the reference's SimpleTokenizer + clip.tokenize(truncate=True) logic on ASCII prompts — the module
    imports ftfy (absent in this image); ftfy.fix_text is the identity on plain ASCII, so the module is loaded with that
    one function standing for it, which is stated here and in the test."""
    import importlib.util
    import types as _types
    spec = importlib.util.spec_from_file_location("ref_clip_model", os.path.join("/root/reference", "CLIP", "clip", "model.py"))
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)
    sd = synth.synth_clip_state_dict(seed=16)
    model = cm.CLIP(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32, context_length=77,
                    vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("attn_mask" in k or k in ("input_resolution", "context_length", "vocab_size") for k in missing), (missing, unexpected)
    img = rnd((2, 3, 224, 224), 1601, 1.0)
    fake = _types.ModuleType("ftfy")
    fake.fix_text = lambda t: t
    sys.modules["ftfy"] = fake
    try:
        spec = importlib.util.spec_from_file_location("ref_clip_tok", os.path.join("/root/reference", "CLIP", "clip", "simple_tokenizer.py"))
        tk = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tk)
    finally:
        del sys.modules["ftfy"]
    tok = tk.SimpleTokenizer()
    prompts = ["a dining chair", "A round table with four legs.", "an L-shaped sofa, grey; 3 seats & 2 cushions", "chair",
               "a very long description " + "of a chair with many many parts " * 12]
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    tokens = torch.zeros(len(prompts), 77, dtype=torch.long)
    for i, t in enumerate(prompts):                      # clip.tokenize(texts, truncate=True), CLIP/clip/clip.py:205-245
        ids = [sot] + tok.encode(t) + [eot]
        if len(ids) > 77:
            ids = ids[:77]
            ids[-1] = eot
        tokens[i, :len(ids)] = torch.tensor(ids)
    with torch.no_grad():
        fi = model.encode_image(img)
        ft = model.encode_text(tokens)
    save("g16_clip_towers", seed=np.array(16), image_seed=np.array(1601), image_features=fi, tokens=tokens, text_features=ft,
         prompts=np.array(prompts))


def g17_spatial_transformer(R):
    """a19 wrappers: the reference's SpatialTransformer / BasicTransformerBlock / GEGLU (modules/attention.py:37-64,
    196-261) on seeded weights (every parameter, incl. the zero-initialised proj_out, drawn from seeded normals) and inputs,
    self- and cross-attention variants."""
    from modules import attention as ratt
    out = {}
    for name, (c, heads, dh, depth, ctx_dim, b, h, w, m) in {"self_d1": (64, 4, 16, 1, None, 2, 6, 5, 0), "cross_d2": (96, 3, 32, 2, 40, 2, 4, 8, 7)}.items():
        mod = ratt.SpatialTransformer(c, heads, dh, depth=depth, context_dim=ctx_dim).eval()
        sd = synth.synth_like({k: tuple(v.shape) for k, v in mod.state_dict().items()}, seed=1700 + len(name), tag="g17")
        mod.load_state_dict(sd, strict=True)
        x = rnd((b, c, h, w), 1750 + len(name))
        ctx = None if ctx_dim is None else rnd((b, m, ctx_dim), 1760 + len(name))
        with torch.no_grad():
            y = mod(x, context=ctx)
        out[name + "__cfg"] = np.array([c, heads, dh, depth, ctx_dim or 0, b, h, w, m])
        out[name + "__x"], out[name + "__y"] = x, y
        if ctx is not None:
            out[name + "__ctx"] = ctx
        out[name + "__weight_seed"] = np.array(1700 + len(name))
    save("g17_spatial_transformer", **out)


XATTN_CASES = [   # name, query_dim, context_dim (None: self), heads, dim_head, b, n, m, masked
    ("self_small", 64, None, 4, 32, 2, 48, 48, False),
    ("cross_ldm", 320, 512, 8, 64, 3, 32, 77, True),            # LDM's usual text-conditioning shape
    ("ragged", 96, 40, 3, 20, 2, 37, 5, True),                 # nothing a multiple of the tile sizes; one sample fully masked
    ("single", 32, 16, 1, 8, 1, 1, 1, False),
    ("long_ctx", 128, 128, 2, 128, 1, 130, 200, False),
]


def g14_cross_attention(R):
    """a19: the reference's CrossAttention module itself on seeded inputs."""
    from modules import attention as ratt
    out = {}
    for name, qd, cd, heads, dh, b, n, m, masked in XATTN_CASES:
        mod = ratt.CrossAttention(qd, cd, heads=heads, dim_head=dh).eval()
        sd = synth.synth_cross_attention_state_dict(qd, cd or qd, heads, dh, seed=len(name))
        mod.load_state_dict(sd, strict=True)
        x = rnd((b, n, qd), 1400 + len(name))
        ctx = None if cd is None else rnd((b, m, cd), 1500 + len(name))
        mask = None
        if masked:
            g = torch.Generator().manual_seed(1600 + len(name))
            mask = torch.rand(b, m, generator=g) > 0.3
            mask[-1] = False                                   # a fully masked sample: uniform attention
        with torch.no_grad():
            y = mod(x, context=ctx, mask=mask)
        out[name + "__cfg"] = np.array([qd, cd or 0, heads, dh, len(name)])      # sizes + the weight seed
        out[name + "__x"] = x
        out[name + "__out"] = y
        if ctx is not None:
            out[name + "__context"] = ctx
        if mask is not None:
            out[name + "__mask"] = mask
    save("g14_cross_attention", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    ap.add_argument("--g10-sizes", default="64,128,256")
    ap.add_argument("--mc512", action="store_true", help="g13: also mesh the 512^3 thin shell (minutes of CPU)")
    a = ap.parse_args()
    torch.manual_seed(0)
    R = import_reference(a.ref)
    jobs = {"g1": lambda: g1_g2(R), "g3": lambda: g3_g4(R), "g5": lambda: g5_g6(R), "g7": lambda: g7_g8(R), "g8flips": lambda: g8_direction_flips(R),
            "g9": lambda: g9(R), "g10": lambda: g10(R, [int(s) for s in a.g10_sizes.split(",")]),
            "g9d64": lambda: g9_d64(R), "g11": lambda: g11_conditioned_loops(R), "g12": lambda: g12_contractive(R),
            "g13": lambda: g13_marching_cubes(R, a.mc512), "g13luts": lambda: g13_lut_hashes(R), "g15": lambda: g15_image_preprocess(R), "g16": lambda: g16_clip_towers(R), "g17": lambda: g17_spatial_transformer(R), "g14": lambda: g14_cross_attention(R)}
    only = [s for s in a.only.split(",") if s]
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        print(f"[{name}]")
        fn()


if __name__ == "__main__":
    main()

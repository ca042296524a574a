"""GPU parity tests (-m gpu): native denoiser + fused reverse loop against the CPU oracle and the
reference-made golden vectors, through the C ABI."""
import types

import numpy as np
import pytest
import torch

from oracle import diffusion as odiff
from oracle import unet as ounet
from surfd_amd import synth
from surfd_amd.spec import UNetConfig

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _args(cond_mode):
    return types.SimpleNamespace(cond_mode=cond_mode, arch="OpenUNet", num_actions=9, dataset="deepfashion3d",
                                 noise_schedule="cosine", sigma_small=True, clip_value=1.0)


_CACHE = {}


def _model(cond_mode, respacing=""):
    from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
    key = (cond_mode, respacing)
    if key not in _CACHE:
        model, diff = create_model_and_diffusion(_args(cond_mode), respacing)
        sd = synth.synth_unet_state_dict(UNetConfig(num_classes=9 if "category" in cond_mode else None))
        load_model_wo_clip(model, sd)
        model.to("cuda")
        model.eval()
        _CACHE[key] = (model, diff, sd)
    return _CACHE[key]


def test_unet_forward_vs_golden(golden):
    model, _, _ = _model("no_cond")
    g = golden("g3_unet_nocond_L32")
    out = model(T(g["x"]).cuda(), T(g["t"]).cuda(), y={})
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)   # stated: abs 1e-4 on O(1) outputs
    model, _, _ = _model("img")
    g = golden("g3_unet_ctx_L64")
    out = model(T(g["x"]).cuda(), T(g["t"]).cuda(), y={"context": T(g["context"]).cuda()})
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)
    model, _, _ = _model("category")
    g = golden("g3_unet_category_L32")
    out = model(T(g["x"]).cuda(), T(g["t"]).cuda(), y={"action_text": T(g["labels"]).cuda()})
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,L", [(1, 32), (3, 32), (8, 32), (5, 64), (8, 64), (2, 8), (16, 16)])
def test_unet_forward_vs_oracle(B, L):
    model, _, sd = _model("no_cond")
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn(B, 1, L, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    out = model(x.cuda(), t.cuda(), y={}).cpu()
    with torch.no_grad():
        ref = ounet.unet_forward(sd, x, t)
    scale = float(ref.abs().max())
    assert float((out - ref).abs().max()) <= 1e-4 * max(1.0, scale)
    # batch independence: each sample alone gives the same result
    one = model(x[:1].cuda(), t[:1].cuda(), y={}).cpu()
    # (different batch sizes tile K differently -> different fp32 summation order, not different maths)
    np.testing.assert_allclose(one.numpy(), out[:1].numpy(), rtol=0, atol=2e-5 * max(1.0, scale))


def test_single_steps_vs_golden(golden):
    model, diff, _ = _model("no_cond")
    g = golden("g5_single_steps")
    x = T(g["x"]).cuda()
    for tt in [999, 500, 1, 0]:
        r = diff.p_sample(model, x, torch.tensor([tt, tt]).cuda(), clip_denoised=False, model_kwargs={"y": {}},
                          _z=T(g[f"z_{tt}"]).cuda())
        np.testing.assert_allclose(r["sample"].cpu().numpy(), g[f"sample_{tt}"], rtol=1e-4, atol=1e-4)
    _, dd, _ = _model("no_cond", "ddim50")
    for tt, eta in [(49, 0.0), (25, 0.0), (0, 0.0), (25, 0.7)]:
        tag = f"{tt}_eta{int(eta * 10)}"
        r = dd.ddim_sample(model, x, torch.tensor([tt, tt]).cuda(), clip_denoised=False, model_kwargs={"y": {}}, eta=eta,
                           _z=T(g[f"ddim_z_{tag}"]).cuda())
        np.testing.assert_allclose(r["sample"].cpu().numpy(), g[f"ddim_sample_{tag}"], rtol=1e-4, atol=1e-4)


def test_step_kernels_bit_exact_vs_oracle():
    """Posterior-update kernels alone (x0 given): same fp32 op order as torch -> equal up to exp ulp."""
    import ctypes as C
    from surfd_amd import _native as N
    L = N.lib()
    g = torch.Generator().manual_seed(4)
    x, x0, z = (torch.randn(8, 1, 32, generator=g) for _ in range(3))
    xd, x0d, zd = x.cuda(), x0.cuda(), z.cuda()        # keep alive: the ABI takes raw pointers
    s = odiff.make_schedule()
    d = odiff.make_schedule(respacing="ddim50")
    for tt in [999, 321, 1, 0]:
        t = torch.full((8,), tt)
        ref = odiff.p_sample(s, lambda a, b: x0, x, t, z)["sample"]
        out = torch.empty(8, 1, 32, device="cuda")
        N.check(L.surfd_ddpm_step(N.ptr(xd), N.ptr(x0d), N.ptr(zd),
                                  float(np.float32(s.posterior_mean_coef1[tt])), float(np.float32(s.posterior_mean_coef2[tt])),
                                  float(np.float32(s.posterior_log_variance_clipped[tt])), int(tt != 0), 0, N.ptr(out), 256,
                                  N.stream()))
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=5e-7)
    for tt, eta in [(49, 0.0), (7, 0.5), (0, 0.0)]:
        t = torch.full((8,), tt)
        ref = odiff.ddim_sample(d, lambda a, b: x0, x, t, z, eta=eta)["sample"]
        out = torch.empty(8, 1, 32, device="cuda")
        f = lambda a: float(np.float32(a[tt]))
        N.check(L.surfd_ddim_step(N.ptr(xd), N.ptr(x0d), N.ptr(zd), f(d.sqrt_recip_alphas_cumprod),
                                  f(d.sqrt_recipm1_alphas_cumprod), f(d.alphas_cumprod), f(d.alphas_cumprod_prev), eta,
                                  int(tt != 0), 0, N.ptr(out), 256, N.stream()))
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_fused_ddim50_vs_golden_and_generic(golden):
    model, _, sd = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim50")
    g = golden("g6_ddim50_B1_L32")
    noise = synth.synth_noise_batch(50, 0, 1, 32, seed=int(g["seed"])).cuda()
    fused = dd.ddim_sample_loop(model, (1, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
    # config C1 end to end.  The synthetic (untrained) denoiser is not contractive, so the 1e-6
    # per-step fp differences grow ~1e2 over 50 steps: stated end-to-end tolerance 1e-3.
    np.testing.assert_allclose(fused.cpu().numpy(), g["x_after_49"], rtol=1e-3, atol=1e-3)
    generic = dd.ddim_sample_loop(model, (1, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=False)
    np.testing.assert_allclose(fused.cpu().numpy(), generic.cpu().numpy(), rtol=1e-3, atol=1e-3)


def test_fused_ddpm1000_loop(golden):
    """The 1000-step ancestral loop (configs C2/C3).  With untrained weights the map x_t -> x_{t-1} is
    chaotic (fp noise doubles every few steps), so end-to-end equality with the reference is not a
    meaningful assertion; instead (a) the first iterations are compared with the golden trajectory,
    (b) every 37th iteration of the fused loop (and the first and last three) is re-derived from its own previous
    state with the generic single-step path (tight per-step tolerance: wrong t / coefficient / noise row would show).  The end-to-end
    comparison with the reference over all 1000 steps is test_ddpm1000_end_to_end_vs_reference (contractive head, G12)."""
    model, diff, _ = _model("no_cond")
    g = golden("g6_ddpm1000_B2_L32")
    noise = synth.synth_noise_batch(1000, 0, 2, 32, seed=int(g["seed"])).cuda()
    out, traj = diff.p_sample_loop(model, (2, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise,
                                   fused=True, return_trajectory=True)
    assert torch.equal(out, traj[-1])
    np.testing.assert_allclose(traj[0].cpu().numpy(), g["x_after_0"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(traj[1].cpu().numpy(), g["x_after_1"], rtol=1e-4, atol=2e-4)
    worst = 0.0
    for k in list(range(0, 1000, 37)) + [1, 2, 997, 998, 999]:
        i = 999 - k
        prev = noise[0] if k == 0 else traj[k - 1]
        r = diff.p_sample(model, prev, torch.tensor([i, i]).cuda(), clip_denoised=False, model_kwargs={"y": {}}, _z=noise[1 + k])
        err = float((r["sample"] - traj[k]).abs().max()) / max(1.0, float(traj[k].abs().max()))
        worst = max(worst, err)
    print(f"fused loop per-step consistency: worst rel err {worst:.2e}")
    assert worst < 5e-5
    assert torch.isfinite(out).all()


def test_cfg_wrapper_is_identity():
    from surfd_amd.mdm import ClassifierFreeSampleModel
    model, diff, _ = _model("img")
    model.cond_mode = "text"            # the wrapper only admits text/action (cfg_sampler.py:21)
    try:
        w = ClassifierFreeSampleModel(model)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(4, 1, 64, generator=g).cuda()
        t = torch.randint(0, 1000, (4,), generator=g).cuda()
        y = {"context": synth.synth_context(0, 4).cuda(), "scale": torch.full((4,), 3.0).cuda()}
        a = w(x, t, y)
        b = model(x, t, y)
        assert torch.equal(a, b)
    finally:
        model.cond_mode = "img"


def test_fused_loop_conditioned_configs():
    """Configs C4/C5 shape (L=64, per-sample 512-d context; C4 through the CFG wrapper) and the category
    mode: the one-call fused loop must agree with the generic per-step Python loop on the same model."""
    from surfd_amd.mdm import ClassifierFreeSampleModel
    _, dd, _ = _model("no_cond", "ddim10")
    model, _, _ = _model("img")
    B, L = 4, 64
    noise = synth.synth_noise_batch(10, 100, B, L).cuda()
    ctx = synth.synth_context(100, B).cuda()
    kw = {"y": {"context": ctx}}
    a = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=True)
    b = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=False)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol=1e-4)
    # a different context must change the result (conditioning really reaches the fused path)
    c = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {"context": ctx.flip(0).contiguous()}},
                            noise_stream=noise, fused=True)
    assert float((a - c).abs().max()) > 1e-3
    # C4: text mode + classifier-free wrapper (scale 3.0)
    model.cond_mode = "text"
    try:
        w = ClassifierFreeSampleModel(model)
        kw4 = {"y": {"context": ctx, "scale": torch.full((B,), 3.0).cuda()}}
        f = dd.ddim_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw4, noise_stream=noise, fused=True)
        g = dd.ddim_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw4, noise_stream=noise, fused=False)
        np.testing.assert_allclose(f.cpu().numpy(), g.cpu().numpy(), rtol=1e-4, atol=1e-4)
        assert torch.equal(f, a)                                   # guidance is an exact no-op in the reference
        with pytest.raises(KeyError):                              # the wrapper needs y['scale'] (cfg_sampler.py:26)
            dd.ddim_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=True)
    finally:
        model.cond_mode = "img"
    cat_model, _, _ = _model("category")
    labels = torch.tensor([0, 3, 8, 5]).cuda()
    noise32 = synth.synth_noise_batch(10, 7, B, 32).cuda()
    kwc = {"y": {"action_text": labels}}
    p = dd.p_sample_loop(cat_model, (B, 1, 32), clip_denoised=True, model_kwargs=kwc, noise_stream=noise32, fused=True)
    q = dd.p_sample_loop(cat_model, (B, 1, 32), clip_denoised=True, model_kwargs=kwc, noise_stream=noise32, fused=False)
    np.testing.assert_allclose(p.cpu().numpy(), q.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_empty_and_single_inputs():
    from surfd_amd.cbndec import CbnDecoder, make_udf_func
    from surfd_amd.spec import DecoderConfig
    dec = CbnDecoder(63, 32, 512, 5)
    dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig()), strict=True)
    dec = dec.cuda().eval()
    lat = torch.zeros(1, 32).cuda()
    f = make_udf_func(dec, lat)
    assert f(torch.zeros(0, 3).cuda()).shape == (0,)
    one = f(torch.tensor([[0.1, -0.2, 0.3]]).cuda())
    assert one.shape == (1,) and 0.0 <= float(one) <= 0.1
    dec.bind_latents(lat)
    u, g = dec.udf_and_ngrad(torch.zeros(0, 3).cuda(), 0)
    assert u.shape == (0,) and g.shape == (0, 3)
    with pytest.raises(RuntimeError, match="not bound"):
        dec.udf(torch.zeros(4, 3).cuda(), 5)                         # latent index outside what was bound


def test_batch_pipeline_matches_sequential():
    """surfd_amd.parallel.BatchPipeline: the next batch's reverse loop on one stream while the previous batch's
    grids are evaluated on another (decoder on part of the CUs) must return the same bits as running the
    batches one after the other."""
    from surfd_amd.cbndec import CbnDecoder, make_udf_func
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.parallel import BatchPipeline
    from surfd_amd.spec import DecoderConfig
    model, _, _ = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim20")
    dec = CbnDecoder(63, 32, 512, 5)
    dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True)
    dec = dec.cuda().eval()
    B, N, nb = 2, 64, 5
    filler = GridFiller(N)
    noise = [synth.synth_noise_batch(20, s * B, B, 32).cuda() for s in range(nb)]
    chains = [model, model.replica()]          # two loops in flight: shared weights, private workspaces / graphs

    def sample(s, chain=0):
        return dd.ddim_sample_loop(chains[chain], (B, 1, 32), clip_denoised=False, model_kwargs={"y": {}},
                                   noise_stream=noise[s], fused=True)

    def make_fill(store):
        def fill(s, lat):
            dec.bind_latents(lat.reshape(B, 32))
            for k in range(B):
                store[(s, k)] = filler.fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16, stats=False)
        return fill

    seq, pip = {}, {}
    for s in range(nb):
        make_fill(seq)(s, sample(s))
    torch.cuda.synchronize()
    BatchPipeline(dec, sample, make_fill(pip), decoder_blocks=96, loop_chains=2).run(nb)
    torch.cuda.synchronize()
    assert set(seq) == set(pip)
    for key in seq:
        assert torch.equal(seq[key][0], pip[key][0]) and torch.equal(seq[key][1], pip[key][1]), key


# ---------------------------------------------------------------------------------------------------------
# round 2: configs C4 / C5 against reference-made trajectories, end-to-end 1000-step equality, per-module
# goldens, the f16x2 / fp32 precision modes, saturation accounting, graph-cache regression
# ---------------------------------------------------------------------------------------------------------
def _model_gain(cond_mode, head_gain):
    from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip
    key = (cond_mode, "gain", head_gain)
    if key not in _CACHE:
        model, diff = create_model_and_diffusion(_args(cond_mode))
        load_model_wo_clip(model, synth.synth_unet_state_dict(head_gain=head_gain))
        model.to("cuda").eval()
        _CACHE[key] = (model, diff)
    return _CACHE[key]


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_c5_image_conditioned_ddim50_vs_reference_trajectory(golden, precision):
    """Config C5's loop (L=64, per-sample 512-d context, B=8) against the trajectory the REFERENCE produced."""
    g = golden("g11_ddim50_img_B8_L64")
    model, _ = _model_gain("img", float(g["head_gain"]))
    _, dd, _ = _model("no_cond", "ddim50")
    model.set_precision(precision)
    try:
        B, L = 8, 64
        noise = synth.synth_noise_batch(50, 0, B, L, seed=int(g["seed"])).cuda()
        ctx = synth.synth_context(0, B, seed=int(g["ctx_seed"])).cuda()
        out = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {"context": ctx}},
                                  noise_stream=noise, fused=True)
        # contractive synthetic head (synth.CONTRACTIVE_HEAD_GAIN): no chaotic amplification, tight end-to-end bound
        np.testing.assert_allclose(out.cpu().numpy(), g["x_after_49"], rtol=0, atol=1e-4)
        assert model.saturation_count() == 0
    finally:
        model.set_precision("f16x2")


def test_c4_text_cfg_ddim50_vs_reference_trajectory(golden):
    """Config C4's loop: text mode under the classifier-free wrapper, scale 3.0 (reference: CLIP tower replaced by a
    fixed embedding table, here passed as y['context'])."""
    from surfd_amd.mdm import ClassifierFreeSampleModel
    g = golden("g11_ddim50_textcfg_B8_L64")
    model, _ = _model_gain("img", float(g["head_gain"]))
    _, dd, _ = _model("no_cond", "ddim50")
    B, L = 8, 64
    noise = synth.synth_noise_batch(50, 0, B, L, seed=int(g["seed"])).cuda()
    ctx = synth.synth_context(0, B, seed=int(g["ctx_seed"])).cuda()
    model.cond_mode = "text"
    try:
        w = ClassifierFreeSampleModel(model)
        kw = {"y": {"context": ctx, "scale": torch.full((B,), float(g["scale"])).cuda()}}
        out = dd.ddim_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=True)
        np.testing.assert_allclose(out.cpu().numpy(), g["x_after_49"], rtol=0, atol=1e-4)
    finally:
        model.cond_mode = "img"


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_ddpm1000_end_to_end_vs_reference(golden, precision):
    """Configs C2/C3's chain asserted END TO END: with the contractive synthetic head the reference's x after all
    1000 ancestral steps is reproduced to 1e-3 (measured ~1e-6), in both conv precisions."""
    g = golden("g12_ddpm1000_contractive_B2_L32")
    model, diff = _model_gain("no_cond", float(g["head_gain"]))
    model.set_precision(precision)
    noise = synth.synth_noise_batch(1000, 0, 2, 32, seed=int(g["seed"])).cuda()
    out, traj = diff.p_sample_loop(model, (2, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise,
                                   fused=True, return_trajectory=True)
    for k in (0, 1, 499, 998, 999):
        err = float(np.abs(traj[k].cpu().numpy() - g[f"x_after_{k}"]).max())
        print(f"{precision}: |x_after_{k} - reference| = {err:.2e}")
        assert err <= 1e-3
    assert float(np.abs(out.cpu().numpy() - g["x_after_999"]).max()) <= 1e-4


def test_c4_ddpm1000_text_cfg_end_to_end_vs_reference(golden):
    """Config C4 at full length: 1000 conditioned steps, L=64, CFG wrapper, contractive head."""
    from surfd_amd.mdm import ClassifierFreeSampleModel, create_model_and_diffusion, load_model_wo_clip
    g = golden("g12_ddpm1000_contractive_textcfg_B2_L64")
    model, diff = create_model_and_diffusion(_args("img"))
    load_model_wo_clip(model, synth.synth_unet_state_dict(head_gain=float(g["head_gain"])))
    model.to("cuda").eval()
    model.cond_mode = "text"
    B, L = 2, 64
    w = ClassifierFreeSampleModel(model)
    noise = synth.synth_noise_batch(1000, 0, B, L, seed=int(g["seed"])).cuda()
    kw = {"y": {"context": synth.synth_context(0, B, seed=int(g["ctx_seed"])).cuda(),
                "scale": torch.full((B,), float(g["scale"])).cuda()}}
    out = diff.p_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=True)
    err = float(np.abs(out.cpu().numpy() - g["x_after_999"]).max())
    print(f"C4 1000-step end-to-end |x - reference| = {err:.2e}")
    assert err <= 1e-4


@pytest.mark.parametrize("design", [80])
def test_wide_form_l64_text_cfg_chain_vs_reference(golden, design):
    """The wide form's 64-position path (two 32-column halves per sample, VEC = 16 kernel) at the bench's design batch against the
    reference's own 1000-step conditioned chain (G12 text + CFG, L = 64): until round 5 only L = 32 chains rode the wide golden
    chain."""
    from surfd_amd.mdm import ClassifierFreeSampleModel, create_model_and_diffusion, load_model_wo_clip
    g = golden("g12_ddpm1000_contractive_textcfg_B2_L64")
    model, diff = create_model_and_diffusion(_args("img"))
    load_model_wo_clip(model, synth.synth_unet_state_dict(head_gain=float(g["head_gain"])))
    model.to("cuda").eval()
    model.cond_mode = "text"
    model.set_wide(design)
    B, L = 2, 64
    w = ClassifierFreeSampleModel(model)
    noise = synth.synth_noise_batch(1000, 0, B, L, seed=int(g["seed"])).cuda()
    kw = {"y": {"context": synth.synth_context(0, B, seed=int(g["ctx_seed"])).cuda(),
                "scale": torch.full((B,), float(g["scale"])).cuda()}}
    out = diff.p_sample_loop(w, (B, 1, L), clip_denoised=False, model_kwargs=kw, noise_stream=noise, fused=True)
    err = float(np.abs(out.cpu().numpy() - g["x_after_999"]).max())
    print(f"wide form (design batch {design}), C4 1000-step end-to-end |x - reference| = {err:.2e}")
    assert err <= 1e-4 and model.saturation_count() == 0


def test_fused_loop_progress_bar_and_time_con(capsys):
    """What every sample/generate_*.py of the reference relies on (gaussian_diffusion.py:677-708): progress=True draws a bar
    over the T iterations, and time_con grows by one entry per iteration.  The fused loop is one C call: the bar is drawn from
    the device-side loop counter (surfd_unet_loop_progress) while the replays are in flight; the result does not depend on it."""
    model, _, _ = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim50")
    B, L, T_ = 2, 32, 50
    noise = synth.synth_noise_batch(T_, 0, B, L).cuda()
    n0 = len(dd.time_con)
    quiet = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True, progress=False).clone()
    assert len(dd.time_con) == n0 + T_
    capsys.readouterr()
    shown = dd.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True, progress=True)
    err = capsys.readouterr().err
    assert torch.equal(shown, quiet)
    assert len(dd.time_con) == n0 + 2 * T_ and all(v >= 0 for v in dd.time_con[n0:])
    assert f"{T_}/{T_}" in err and "100%" in err, err[-300:]
    from surfd_amd import _native as N
    import ctypes as C
    Lh, h = model._native()
    it = C.c_int(-7)
    N.check(Lh.surfd_unet_loop_progress(h, C.byref(it)))
    assert it.value == T_                                   # the counter the bar read: every iteration of the last loop


def test_unet_precision_modes_agree():
    """f16x2 (default) against the exact fp32 conv kernels on the same inputs: fp32-class agreement."""
    model, _, sd = _model("no_cond")
    g = torch.Generator().manual_seed(77)
    for B, L in [(8, 32), (3, 64), (16, 16)]:
        x = torch.randn(B, 1, L, generator=g).cuda()
        t = torch.randint(0, 1000, (B,), generator=g).cuda()
        model.set_precision("fp32")
        a = model(x, t, y={}).clone()
        model.set_precision("f16x2")
        b = model(x, t, y={})
        scale = max(1.0, float(a.abs().max()))
        assert float((a - b).abs().max()) <= 2e-5 * scale
    assert model.saturation_count() == 0


def test_cu_budget_changes_summation_order_only():
    """MDM.set_cu_budget sizes the split-K of the convs (a loop that shares the chip gets fewer, larger workgroups): the
    forward agrees with the default to fp32 rounding, a cached loop graph is re-captured, and restoring the budget
    restores the bits."""
    model, _, _ = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim10")
    g = torch.Generator().manual_seed(78)
    x = torch.randn(8, 1, 32, generator=g).cuda()
    t = torch.randint(0, 1000, (8,), generator=g).cuda()
    noise = synth.synth_noise_batch(10, 0, 8, 32).cuda()
    run = lambda: dd.ddim_sample_loop(model, (8, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True).clone()
    a, la = model(x, t, y={}).clone(), run()
    try:
        for cus in (16, 64):
            model.set_cu_budget(cus)
            b, lb = model(x, t, y={}).clone(), run()
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))
            assert float((la - lb).abs().max()) <= 1e-4
    finally:
        model.set_cu_budget(256)
    assert torch.equal(model(x, t, y={}), a) and torch.equal(run(), la)
    with pytest.raises(RuntimeError):
        model.set_cu_budget(0)


def test_unet_saturation_is_counted():
    """Operands beyond the fp16 range are clamped by the f16x2 convs and MUST be reported (never silent)."""
    model, _, _ = _model("no_cond")
    model.saturation_count()
    x = torch.full((2, 1, 32), 1.0e6).cuda()           # the first conv stages x itself (no GroupNorm in front of it)
    model(x, torch.tensor([5, 5]).cuda(), y={})
    assert model.saturation_count() > 0
    assert model.saturation_count() == 0               # reset by the read
    model.set_precision("fp32")
    try:
        out = model(x, torch.tensor([5, 5]).cuda(), y={})
        assert torch.isfinite(out).all() and model.saturation_count() == 0
    finally:
        model.set_precision("f16x2")


def test_range_guard_on_the_denoiser(capsys):
    """surfd_amd.rangeguard with the denoiser's counter: an input beyond the fp16 range makes the first conv clamp; the guard
    runs the evaluation again in the exact-fp32 mode and returns that result."""
    from surfd_amd.rangeguard import run_guarded
    model, _, _ = _model("no_cond")
    x = torch.full((2, 1, 32), 1.0e6).cuda()
    t = torch.tensor([5, 5]).cuda()
    try:
        out, clamped = run_guarded("reverse loop (denoiser)", lambda: model(x, t, y={}), model.saturation_count, lambda: model.set_precision("fp32"))
        assert clamped > 0 and "reverse loop (denoiser)" in capsys.readouterr().err
        assert torch.isfinite(out).all() and torch.equal(out, model(x, t, y={})) and model.saturation_count() == 0
    finally:
        model.set_precision("f16x2")


def test_interleaved_loops_match_separate_loops():
    """several loops driven by ONE host thread (surfd_sample_loop_begin / _run / _end, `chunk` graph replays per
    loop in turn) give, loop for loop, the bits of one surfd_sample_loop call each — latency form and wide form, conditioned and
    not, any chunk; and the three calls refuse to be used out of order."""
    import ctypes as C
    from surfd_amd import _native as N
    model, _, _ = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim20")
    chains = [model, model.replica(), model.replica()]
    streams = [torch.cuda.Stream() for _ in chains]
    try:
        for wide, widths in ((0, (8, 3, 5)), (40, (40, 40, 27))):
            for m in chains:
                m.set_wide(wide)
            noise = [synth.synth_noise_batch(20, 100 * q, w, 32).cuda() for q, w in enumerate(widths)]
            ref = [dd.p_sample_loop(chains[q], (w, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise[q], fused=True).clone()
                   for q, w in enumerate(widths)]
            torch.cuda.synchronize()
            for chunk in (1, 3, 20, 64):
                jobs = [{"model": chains[q], "shape": (w, 1, 32), "noise_stream": noise[q], "stream": streams[q], "model_kwargs": {"y": {}}}
                        for q, w in enumerate(widths)]
                n0 = len(dd.time_con)
                outs = dd.fused_loops_interleaved(jobs, sampler="ddpm", clip_denoised=False, chunk=chunk)
                torch.cuda.synchronize()
                assert all(torch.equal(a, b) for a, b in zip(ref, outs)), (wide, chunk)
                assert len(dd.time_con) == n0 + 20
        with pytest.raises(ValueError):
            dd.fused_loops_interleaved([{"model": model, "shape": (2, 1, 32), "noise_stream": noise[0][:, :2].contiguous(), "stream": streams[q]} for q in range(2)])
    finally:
        for m in chains:
            m.set_wide(0)
    # protocol: run / end need an open loop; end needs every iteration launched
    fresh = model.replica()
    Lh, h = fresh._native()
    out = torch.empty(2, 1, 32, device="cuda")
    rem = C.c_int(0)
    for call in (lambda: Lh.surfd_sample_loop_run(h, 1, C.byref(rem), N.stream()), lambda: Lh.surfd_sample_loop_end(h, N.ptr(out), N.stream())):
        with pytest.raises(RuntimeError):
            N.check(call())
    a = dd._fused_inputs(fresh, (2, 1, 32), "ddpm", None, synth.synth_noise_batch(20, 0, 2, 32).cuda(), False, {"y": {}}, 0.0, torch.device("cuda"), False)
    N.check(Lh.surfd_sample_loop_begin(h, C.byref(a["cfg"]), N.ptr(a["noise"]), None, None, None, 2, 32, N.stream()))
    N.check(Lh.surfd_sample_loop_run(h, 7, C.byref(rem), N.stream()))
    assert rem.value == 13
    with pytest.raises(RuntimeError, match="7 of 20"):
        N.check(Lh.surfd_sample_loop_end(h, N.ptr(out), N.stream()))
    N.check(Lh.surfd_sample_loop_run(h, 1000, C.byref(rem), N.stream()))          # clipped to what is left
    assert rem.value == 0
    N.check(Lh.surfd_sample_loop_end(h, N.ptr(out), N.stream()))
    torch.cuda.synchronize()
    assert torch.equal(out, dd.p_sample_loop(model, (2, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=a["noise"], fused=True))


def test_cached_loop_graph_survives_workspace_growth():
    """the cached loop graph bakes workspace / embedding-table pointers in; a larger-B call in
    between reallocates them.  The loop must re-capture, not replay against freed memory."""
    model, _, _ = _model("no_cond")
    _, dd, _ = _model("no_cond", "ddim10")
    noise = synth.synth_noise_batch(10, 0, 2, 32).cuda()
    run = lambda: dd.ddim_sample_loop(model, (2, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
    a = run().clone()
    big = synth.synth_noise_batch(10, 50, 24, 64).cuda()          # grows B, L and the embedding table
    dd.ddim_sample_loop(model, (24, 1, 64), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=big, fused=True)
    model(torch.randn(40, 1, 64).cuda(), torch.randint(0, 1000, (40,)).cuda(), y={})
    b = run()
    assert torch.equal(a, b)
    _, d1000, _ = _model("no_cond")
    n1000 = synth.synth_noise_batch(1000, 0, 1, 32).cuda()        # T = 1000 rows: table reallocated again
    d1000.p_sample_loop(model, (1, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=n1000, fused=True)
    assert torch.equal(a, run())


G4_CASES = [("input_blocks__1__0", "res"), ("input_blocks__4__0", "res"), ("output_blocks__0__0", "res"),
            ("output_blocks__5__0", "res"), ("input_blocks__1__1", "attn"), ("middle_block__1", "attn"),
            ("output_blocks__2__1", "attn"), ("input_blocks__3__0", "down"), ("out", "head")]


def test_per_module_activations_vs_golden(golden):
    """a9-a12 per module on the GPU: each module's ops alone, fed the input activation the REFERENCE's forward hook
    recorded (G4), must reproduce the hooked output (ResBlocks incl. skip conv, AttentionBlocks at three widths,
    Downsample, head)."""
    from surfd_amd import _native as N
    g4 = golden("g4_modules_nocond_L32")
    g3 = golden("g3_unet_nocond_L32")
    model, _, _ = _model("no_cond")
    L, h = model._native()
    try:
        for precision in ("fp32", "f16x2"):
            model.set_precision(precision)
            x, t = T(g3["x"]).cuda(), T(g3["t"]).cuda()
            model(x, t, y={})                       # prepares the embedding rows of (t) and the workspace
            for name, kind in G4_CASES:
                xin = T(g4[name + "__in"]).cuda().contiguous()
                ref = g4[name + "__out"]
                out = torch.empty(ref.shape, device="cuda")
                N.check(L.surfd_unet_debug_run_module(h, name.replace("__", ".").encode(), N.ptr(xin), xin.shape[1], xin.shape[2],
                                                      N.ptr(out), ref.shape[1], ref.shape[2], xin.shape[0], 32, N.stream()))
                err = float(np.abs(out.cpu().numpy() - ref).max())
                scale = max(1.0, float(np.abs(ref).max()))
                assert err <= 2e-5 * scale, (precision, name, kind, err)
    finally:
        model.set_precision("f16x2")


# ---- wide form of the conv kernel (MDM.set_wide): four row tiles per workgroup, batch-independent K split ----
@pytest.fixture
def wide_model(request):
    """design batch 32 unless a test asks for another one (indirect parameter): 80 is what bench.py rides"""
    model, diff, sd = _model("no_cond")
    model.set_wide(getattr(request, "param", 32))
    try:
        yield model, diff, sd
    finally:
        model.set_wide(0)


@pytest.mark.parametrize("wide_model", [32, 160], indirect=True)
@pytest.mark.parametrize("B,L", [(1, 32), (8, 32), (5, 64), (2, 8), (16, 16), (40, 32), (80, 32), (80, 64), (160, 32), (3, 16), (33, 8)])   # 80 = the width the bench rides
def test_wide_form_forward_vs_oracle(wide_model, B, L):
    model, _, sd = wide_model
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn(B, 1, L, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    xc, tc = x.cuda(), t.cuda()
    dev = model(xc, tc, y={}).clone()
    out = dev.cpu()
    # EVERY sample (the oracle below sees five): bit-stable from run to run and within the bound of the exact-fp32 mode.  Round 5:
    # a build whose second workgroup per CU went wrong at L = 64 (1e-3, different every run) was caught here only because sample
    # B // 2 happened to be one of the faulty ones
    assert torch.equal(dev, model(xc, tc, y={}))
    model.set_precision("fp32")
    try:
        exact = model(xc, tc, y={}).clone()
    finally:
        model.set_precision("f16x2")
    assert float((dev - exact).abs().max()) <= 1e-4
    # the oracle takes seconds per sample; samples are independent: the first two, the last two (the ragged last workgroup
    # of every launch) and one in the middle
    pick = sorted(set([0, 1, B // 2, B - 2, B - 1]) & set(range(B)))
    with torch.no_grad():
        ref = ounet.unet_forward(sd, x[pick], t[pick])
    assert float((out[pick] - ref).abs().max()) <= 1e-4
    assert model.saturation_count() == 0


@pytest.mark.parametrize("wide_model", [32, 80, 160], indirect=True)      # 160: two column tiles per wave on 112-channel K blocks (NT2)
def test_wide_form_vs_golden_modules_and_chain(wide_model, golden):
    """The wide form against the same reference-made fixtures as the latency form: whole forward (G3), per-module
    activations (G4) and the 1000-step contractive chain end to end (G12)."""
    from surfd_amd import _native as N
    model, _, _ = wide_model
    g3, g4 = golden("g3_unet_nocond_L32"), golden("g4_modules_nocond_L32")
    x, t = T(g3["x"]).cuda(), T(g3["t"]).cuda()
    np.testing.assert_allclose(model(x, t, y={}).cpu().numpy(), g3["out"], rtol=1e-4, atol=1e-4)
    L, h = model._native()
    for name, kind in G4_CASES:
        xin = T(g4[name + "__in"]).cuda().contiguous()
        ref = g4[name + "__out"]
        out = torch.empty(ref.shape, device="cuda")
        N.check(L.surfd_unet_debug_run_module(h, name.replace("__", ".").encode(), N.ptr(xin), xin.shape[1], xin.shape[2],
                                              N.ptr(out), ref.shape[1], ref.shape[2], xin.shape[0], 32, N.stream()))
        err = float(np.abs(out.cpu().numpy() - ref).max())
        assert err <= 2e-5 * max(1.0, float(np.abs(ref).max())), (name, kind, err)
    g = golden("g12_ddpm1000_contractive_B2_L32")
    cm, diff = _model_gain("no_cond", float(g["head_gain"]))
    cm.set_wide(32)
    try:
        noise = synth.synth_noise_batch(1000, 0, 2, 32, seed=int(g["seed"])).cuda()
        out = diff.p_sample_loop(cm, (2, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
        assert float(np.abs(out.cpu().numpy() - g["x_after_999"]).max()) <= 1e-4
    finally:
        cm.set_wide(0)


@pytest.mark.parametrize("wide_model", [32, 160], indirect=True)
def test_wide_form_latent_does_not_depend_on_batch_width(wide_model):
    """A shape's latent is bit-identical whatever loop batch it rode in (noise is seeded per global shape index): the wide
    form's K split is a function of the layer, not of B.  40 DDIM steps amplify any last-bit difference."""
    model, _, _ = wide_model
    _, dd, _ = _model("no_cond", "ddim40")
    run = lambda first, n: dd.ddim_sample_loop(model, (n, 1, 32), clip_denoised=False, model_kwargs={"y": {}},
                                               noise_stream=synth.synth_noise_batch(40, first, n, 32).cuda(), fused=True).clone()
    whole = run(0, 40)
    for first, n in [(0, 1), (0, 8), (8, 8), (16, 24), (37, 3)]:
        assert torch.equal(run(first, n), whole[first:first + n]), (first, n)
    x = torch.randn(40, 1, 32, generator=torch.Generator().manual_seed(5)).cuda()
    t = torch.randint(0, 1000, (40,), generator=torch.Generator().manual_seed(6)).cuda()
    a = model(x, t, y={}).clone()
    assert torch.equal(model(x[11:14].contiguous(), t[11:14].contiguous(), y={}), a[11:14])
    model.set_wide(0)                                # the latency form agrees to fp32 rounding
    b = model(x, t, y={})
    assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))


def test_phased_pipeline_matches_sequential(wide_model):
    """surfd_amd.parallel.PhasedPipeline (bench.py's default schedule): the reverse loops of a round as two WIDE loops over
    several steps' latents at once, then the round's grids — every shape's latent, grid and gradients must equal, bit for
    bit, what one narrow loop per step followed by its grids gives (same conv form), for round sizes that do not divide the
    number of steps."""
    from surfd_amd.cbndec import CbnDecoder, make_udf_func
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.parallel import PhasedPipeline
    from surfd_amd.spec import DecoderConfig
    model, _, _ = wide_model
    _, dd, _ = _model("no_cond", "ddim20")
    dec = CbnDecoder(63, 32, 512, 5)
    dec.load_state_dict(synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32)), strict=True)
    dec = dec.cuda().eval()
    B, N, nb = 2, 64, 7
    filler = GridFiller(N)
    bank = torch.stack([synth.synth_noise_batch(20, s * B, B, 32) for s in range(nb)], 0).cuda()        # [S, T+1, B, 1, L]
    rep = model.replica()
    rep.set_wide(32)
    chains = [model, rep]

    def loop(first, n, chain=0):
        noise = bank[first:first + n].permute(1, 0, 2, 3, 4).reshape(21, n * B, 1, 32).contiguous()
        return dd.ddim_sample_loop(chains[chain], (n * B, 1, 32), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)

    def make_fill(store):
        def fill(s, lat):
            dec.bind_latents(lat.reshape(B, 32))
            for k in range(B):
                u, g = filler.fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16, stats=False)
                store[(s, k)] = (lat[k].clone(), u.clone(), g.clone())
        return fill

    seq, pip = {}, {}
    for s in range(nb):
        make_fill(seq)(s, loop(s, 1))
    torch.cuda.synchronize()
    pipe = PhasedPipeline(loop, make_fill(pip), chains=2, max_loop_batches=2)          # rounds of 4, 3 steps: loops of 2+2 and 2+1 steps
    assert [len(p) for _, p in pipe.plan(nb)] == [2, 2] and sum(n for _, p in pipe.plan(nb) for _, _, n in p) == nb
    pipe.record_timeline = True
    pipe.run(nb)
    torch.cuda.synchronize()
    assert set(seq) == set(pip) and len(pipe.timeline) == 2
    for key in seq:
        for a, b in zip(seq[key], pip[key]):
            assert torch.equal(a, b), key
    assert all(t["grids_done_ms"] >= t["loops_done_ms"] > 0 for t in pipe.timeline)
    # the overlapped variant (next round's loops beside this round's grids on part of the CUs, smaller first round): same bits
    ovl = {}
    pipe = PhasedPipeline(loop, make_fill(ovl), chains=2, max_loop_batches=2, overlap_blocks=96, decoder=dec, first_round_batches=1)
    assert [sum(n for _, _, n in p) for _, p in pipe.plan(nb)] == [1, 3, 3]
    pipe.run(nb)
    torch.cuda.synchronize()
    assert set(ovl) == set(seq)
    for key in seq:
        for a, b in zip(seq[key], ovl[key]):
            assert torch.equal(a, b), key
    # round 5: the loops of a round driven by ONE host thread (graph replays handed to the loops in turn) — both schedules, same bits
    def loops(parts, streams):
        jobs = []
        for (c, first, n), st in zip(parts, streams):
            with torch.cuda.stream(st):
                jobs.append({"model": chains[c], "shape": (n * B, 1, 32), "stream": st, "model_kwargs": {"y": {}},
                             "noise_stream": bank[first:first + n].permute(1, 0, 2, 3, 4).reshape(21, n * B, 1, 32).contiguous()})
        return dd.fused_loops_interleaved(jobs, sampler="ddim", clip_denoised=False, chunk=2, wait_current=False)

    for kw in ({}, {"overlap_blocks": 96, "decoder": dec, "first_round_batches": 1}):
        one = {}
        pipe = PhasedPipeline(loop, make_fill(one), chains=2, max_loop_batches=2, loops_fn=loops, **kw)
        pipe.run(nb)
        torch.cuda.synchronize()
        assert set(one) == set(seq)
        for key in seq:
            for a, b in zip(seq[key], one[key]):
                assert torch.equal(a, b), (key, kw.keys())
    with pytest.raises(ValueError):
        PhasedPipeline(loop, make_fill({}), overlap_blocks=96)                    # needs the decoder it sizes


def test_wide_form_conditioned_latent_does_not_depend_on_batch_width():
    """The same contract with per-sample conditioning (configs C4 / C5: a 512-d context row per latent, L = 64): a short
    conditioned DDIM loop over 12 latents equals, bit for bit, the loops over 5 + 7 of them, and the forward of a sub-batch
    equals the rows of the full batch (the embedding Linears chunk their rows and K axis independently of B in this form)."""
    model, _, _ = _model("img")
    _, dd, _ = _model("no_cond", "ddim10")
    model.set_wide(32)
    try:
        L = 64
        ctx = synth.synth_context(300, 12).cuda()
        run = lambda first, n: dd.ddim_sample_loop(model, (n, 1, L), clip_denoised=False, model_kwargs={"y": {"context": ctx[first:first + n].contiguous()}},
                                                   noise_stream=synth.synth_noise_batch(10, 300 + first, n, L).cuda(), fused=True).clone()
        whole = run(0, 12)
        assert torch.equal(run(0, 5), whole[:5]) and torch.equal(run(5, 7), whole[5:])
        x = torch.randn(12, 1, L, generator=torch.Generator().manual_seed(9)).cuda()
        t = torch.randint(0, 1000, (12,), generator=torch.Generator().manual_seed(10)).cuda()
        full = model(x, t, y={"context": ctx}).clone()
        part = model(x[3:6].contiguous(), t[3:6].contiguous(), y={"context": ctx[3:6].contiguous()})
        assert torch.equal(part, full[3:6])
        # the widths the bench rides: one conditioned L = 64 loop of 80 latents against loops of 8 and 40 of them —
        # every 64-position layer runs as two 32-column halves of the wide kernel, the K split fixed per layer
        ctx = synth.synth_context(500, 80).cuda()
        run = lambda first, n: dd.ddim_sample_loop(model, (n, 1, L), clip_denoised=False, model_kwargs={"y": {"context": ctx[first:first + n].contiguous()}},
                                                   noise_stream=synth.synth_noise_batch(10, 500 + first, n, L).cuda(), fused=True).clone()
        whole = run(0, 80)
        assert torch.equal(run(0, 8), whole[:8]) and torch.equal(run(40, 40), whole[40:]) and torch.equal(run(71, 9), whole[71:])
        assert torch.isfinite(whole).all() and model.saturation_count() == 0
    finally:
        model.set_wide(0)


# ---- run-to-run bit stability of every instantiation -------------------------------------------------------
# Round 5's two instabilities (profiles/r06_conv2_instability.md) gave a DIFFERENT wrong result in most evaluations: 40 evaluations
# of the same input through the same handle see them with certainty, and cost a fraction of a second per case.
_STABILITY_CASES = [
    # (B, L, wide design batch): which conv2_kernel instantiations an evaluation launches
    (8, 32, 0),       # latency form, VEC = 8 (what sample/generate_* runs)
    (8, 64, 0),       # latency form, VEC = 16 at the 64-position level
    (80, 32, 80),     # lean wide form: three workgroups per CU (the bench's loops)
    (53, 32, 80),     # the same with a ragged last batch chunk
    (160, 32, 160),   # two column tiles per wave (NT2)
    (80, 64, 32),     # VEC = 16 wide form (two workgroups per CU) at the 64-position level + lean form below it: round 5's failing case
    (40, 64, 80),
    (24, 64, 32),     # the smallest batch at which round 5's build went wrong
]


@pytest.mark.parametrize("B,L,wide", _STABILITY_CASES)
def test_denoiser_evaluation_is_bit_stable_run_to_run(B, L, wide):
    model, _, _ = _model("no_cond")
    model.set_wide(wide)
    try:
        g = torch.Generator().manual_seed(7 * B + L)
        x = torch.randn(B, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
        first = model(x, t, y={}).clone()
        for run in range(1, 40):
            out = model(x, t, y={})
            if not torch.equal(out, first):
                d = (out - first).abs()
                bad = sorted(set(torch.nonzero(d.flatten(1).sum(1)).flatten().tolist()))
                raise AssertionError(f"evaluation {run} differs from evaluation 0: max |d| = {float(d.max()):.3e} in samples {bad[:16]}")
        # and it is the right result: every sample against the exact-fp32 kernels
        model.set_precision("fp32")
        try:
            exact = model(x, t, y={}).clone()
        finally:
            model.set_precision("f16x2")
        assert float((first - exact).abs().max()) <= 1e-4
        assert model.saturation_count() == 0
    finally:
        model.set_wide(0)


@pytest.mark.parametrize("B,L,wide", [(8, 32, 0), (80, 32, 80), (8, 64, 0), (80, 64, 80)])
def test_fused_loop_is_bit_stable_run_to_run(B, L, wide):
    """The head convolution inside the graph-replayed loop (the LF instantiations: posterior update in the epilogue) and 20
    consecutive evaluations feeding each other: the same noise gives the same latents, 8 times."""
    model, diff, _ = _model("no_cond", "ddim20")
    model.set_wide(wide)
    try:
        noise = synth.synth_noise_batch(diff.num_timesteps, 0, B, L).cuda()
        first = diff.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True).clone()
        for run in range(1, 8):
            out = diff.ddim_sample_loop(model, (B, 1, L), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, fused=True)
            assert torch.equal(out, first), f"loop {run}: max |d| = {float((out - first).abs().max()):.3e}"
        assert torch.isfinite(first).all()
    finally:
        model.set_wide(0)

"""CPU tests of the host-side mirror of the reference interface: schedules, respacing, the generic
reverse loops (any model callable), module state_dict layouts and loaders."""
import types

import numpy as np
import pytest
import torch

from oracle import diffusion as odiff
from surfd_amd import synth
from surfd_amd.diffusion import SpacedDiffusion, create_gaussian_diffusion, get_named_beta_schedule, space_timesteps
from surfd_amd.spec import DecoderConfig, UNetConfig, decoder_param_spec, unet_param_spec

ARGS = types.SimpleNamespace(cond_mode="no_cond", arch="OpenUNet", num_actions=9, dataset="d", noise_schedule="cosine",
                             sigma_small=True, clip_value=1.0)


def toy_model(x, t, **kw):
    # any nonlinear, timestep-dependent map will do: exercises the timestep map and coefficient tables
    return torch.tanh(x * 0.7 + 0.3) * (1.0 + t.view(-1, 1, 1).float() / 1000.0)


class ToyModule(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))

    def forward(self, x, t, **kw):
        return toy_model(x, t)


def test_schedule_tables_match_golden(golden):
    g = golden("g2_schedule")
    d = create_gaussian_diffusion(ARGS)
    for n in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]:
        np.testing.assert_array_equal(getattr(d, n), g[n], err_msg=n)
    assert d.timestep_map == list(g["full_timestep_map"])
    dd = create_gaussian_diffusion(ARGS, "ddim50")
    assert dd.timestep_map == list(g["ddim50_timestep_map"]) and dd.num_timesteps == 50
    np.testing.assert_array_equal(dd.posterior_mean_coef1, g["ddim50_posterior_mean_coef1"])
    np.testing.assert_array_equal(get_named_beta_schedule("cosine", 1000), g["base_betas"])
    assert sorted(space_timesteps(300, [10, 15, 20])) == list(g["sections_10_15_20_of_300"])
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        space_timesteps(10, [20])


@pytest.mark.parametrize("sampler,resp,eta", [("ddpm", "", 0.0), ("ddpm", "100", 0.0), ("ddim", "ddim50", 0.0), ("ddim", "ddim20", 0.5)])
def test_generic_loops_equal_oracle(sampler, resp, eta):
    d = create_gaussian_diffusion(ARGS, resp)
    s = odiff.make_schedule("cosine", 1000, resp)
    T = d.num_timesteps
    assert T == s.num_timesteps
    noise = synth.synth_noise_batch(T, 0, 3, 16)
    m = ToyModule()
    if sampler == "ddpm":
        out = d.p_sample_loop(m, (3, 1, 16), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise)
    else:
        out = d.ddim_sample_loop(m, (3, 1, 16), clip_denoised=False, model_kwargs={"y": {}}, noise_stream=noise, eta=eta)
    ref = odiff.sample_loop(s, lambda x, t: toy_model(x, t), noise, sampler=sampler, eta=eta)
    assert torch.equal(out, ref)
    assert len(d.time_con) == T


def test_loop_contract_details():
    d = create_gaussian_diffusion(ARGS, "ddim10")
    m = ToyModule()
    with pytest.raises(KeyError):                      # model_kwargs must carry a dict 'y' (gaussian_diffusion.py:288)
        d.p_sample_loop(m, (1, 1, 8), model_kwargs={})
    dumps = d.p_sample_loop(m, (2, 1, 8), clip_denoised=True, model_kwargs={"y": {}}, dump_steps=[0, 9])
    assert len(dumps) == 2 and dumps[0].shape == (2, 1, 8)
    x = d.p_sample_loop(m, (2, 1, 8), model_kwargs={"y": {}}, skip_timesteps=4, init_image=torch.ones(2, 1, 8))
    assert x.shape == (2, 1, 8) and torch.isfinite(x).all()
    with pytest.raises(RuntimeError):
        d.p_sample_loop(m, (1, 1, 8), model_kwargs={"y": {}}, fused=True)     # fused needs the native model on a GPU
    # device discovery from the model's parameters when device=None
    assert d.p_sample_loop(m, (1, 1, 8), model_kwargs={"y": {}}).device.type == "cpu"


def test_module_layouts_and_loaders():
    from surfd_amd.cbndec import CbnDecoder, CoordsEncoder
    from surfd_amd.mdm import MDM, ClassifierFreeSampleModel, create_model_and_diffusion, load_model_wo_clip
    small = UNetConfig(channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=(2,))
    m = MDM(cond_mode="no_cond", unet_cfg=small)
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == unet_param_spec(small)
    sd = synth.synth_unet_state_dict(small)
    load_model_wo_clip(m, sd)
    assert torch.equal(m.state_dict()["Unet.out.2.weight"], sd["Unet.out.2.weight"])
    with pytest.raises(AssertionError):
        bad = dict(sd)
        bad.pop("Unet.out.2.bias")
        load_model_wo_clip(m, bad)
    assert m.eval() is None or True                       # reference's train() override returns None
    dec = CbnDecoder(63, 32, 512, 5)
    assert [(k, tuple(v.shape)) for k, v in dec.state_dict().items()] == decoder_param_spec(DecoderConfig())
    dec.load_state_dict(synth.synth_decoder_state_dict(), strict=True)
    with pytest.raises(RuntimeError):
        dec.load_state_dict({"decoder.fc_p.weight": torch.zeros(512, 63, 1)}, strict=True)
    enc = CoordsEncoder()
    assert enc.out_dim == 63 and enc.encode(torch.zeros(2, 5, 3)).shape == (2, 5, 63)
    w = ClassifierFreeSampleModel(m)
    with pytest.raises(AssertionError):                   # only text/action models may be wrapped (cfg_sampler.py:21)
        w(torch.zeros(1, 1, 32), torch.zeros(1, dtype=torch.long), y={"scale": torch.ones(1)})
    model, diff = create_model_and_diffusion(types.SimpleNamespace(**{**vars(ARGS), "cond_mode": "category"}))
    assert "Unet.label_emb.weight" in model.state_dict() and diff.num_timesteps == 1000


def test_gridfiller_levels_and_sharding():
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.parallel import shard_range
    assert GridFiller(64).N_levels == [32, 64] and GridFiller(512).N_levels == [32, 64, 128, 256, 512]
    for total, world in [(64, 8), (10, 4), (3, 8), (0, 2)]:
        blocks = [shard_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in blocks) == total
        pos = 0
        for first, count in blocks:
            assert first == pos
            pos += count
    # noise is a function of the global shape index only
    a = synth.synth_noise_batch(5, 0, 4, 8)
    b = synth.synth_noise_batch(5, 2, 2, 8)
    assert torch.equal(a[:, 2:], b)


def test_decoder_state_key_sees_rebound_tensors():
    """the native copy of the decoder weights is refreshed whenever the key changes; the key must change when a
    parameter OBJECT is replaced without going through .to() / load_state_dict (attribute assignment, assign=True loads)."""
    import torch
    from surfd_amd.cbndec import CbnDecoder
    dec = CbnDecoder(63, 32, 512, 5)
    k0 = dec._state_key()
    assert k0 == dec._state_key() and len(k0) == len(dec.state_dict())
    owner = getattr(dec.decoder.blocks, "0").fc_0
    owner.weight = torch.nn.Parameter(owner.weight.detach().clone())          # same values, new object
    k1 = dec._state_key()
    assert k1 != k0
    with torch.no_grad():
        owner.weight.add_(1.0)                                      # in-place: version bump
    assert dec._state_key() != k1
    sd = {k: v.clone() for k, v in dec.state_dict().items()}
    k2 = dec._state_key()
    dec.load_state_dict(sd, assign=True)
    assert dec._state_key() != k2


def test_phased_pipeline_plan_covers_every_batch_once():
    """parallel.PhasedPipeline.plan: rounds of at most chains x max_loop_batches batches, every batch in exactly one loop,
    loops of a round balanced to within one batch, never an empty loop."""
    from surfd_amd.parallel import PhasedPipeline
    for chains, mx in [(1, 4), (2, 10), (2, 2), (3, 5)]:
        pipe = PhasedPipeline(None, None, chains=chains, max_loop_batches=mx)
        for n in list(range(1, 30)) + [40, 97]:
            seen = []
            for first, parts in pipe.plan(n):
                sizes = [cnt for _, _, cnt in parts]
                assert 1 <= len(parts) <= chains and min(sizes) >= 1 and max(sizes) <= mx and max(sizes) - min(sizes) <= 1
                assert parts[0][1] == first and len({c for c, _, _ in parts}) == len(parts)
                for _, f, cnt in parts:
                    seen.extend(range(f, f + cnt))
            assert seen == list(range(n)), (chains, mx, n)
    assert PhasedPipeline(None, None, chains=2, max_loop_batches=10).plan(20) == [(0, [(0, 0, 10), (1, 10, 10)])]      # the driver command: one round


def test_range_guard_control_flow():
    """surfd_amd.rangeguard.run_guarded (what examples/generate.py wraps the reverse loop and every shape's grids in): a clean
    stage runs once; a stage whose saturation counter is non-zero is run again after the handle has been switched to exact
    fp32, with one line logged; --strict raises instead; counts left behind by an earlier stage are not this stage's."""
    from surfd_amd.rangeguard import RangeError, run_guarded

    class Handle:
        def __init__(self, clamps_in_f16x2, stale=0):
            self.mode, self.pending, self.runs, self.clamps = "f16x2", stale, [], clamps_in_f16x2

        def run(self):
            self.runs.append(self.mode)
            if self.mode == "f16x2":
                self.pending += self.clamps
            return f"result[{self.mode}]"

        def read(self):                      # reads AND resets, like surfd_*_saturation_count(reset=1)
            n, self.pending = self.pending, 0
            return n

        def to_fp32(self):
            self.mode = "fp32"

    lines = []
    h = Handle(0, stale=7)                   # an earlier stage's count must not trigger a re-run
    assert run_guarded("stage", h.run, h.read, h.to_fp32, log=lines.append) == ("result[f16x2]", 0)
    assert h.runs == ["f16x2"] and lines == []
    h = Handle(3)
    out, clamped = run_guarded("shape 2 (decoder grids)", h.run, h.read, h.to_fp32, log=lines.append)
    assert out == "result[fp32]" and clamped == 3 and h.runs == ["f16x2", "fp32"] and h.mode == "fp32"
    assert len(lines) == 1 and "shape 2 (decoder grids)" in lines[0] and "3 workgroup" in lines[0] and "fp32" in lines[0]
    out, clamped = run_guarded("shape 3", h.run, h.read, h.to_fp32, log=lines.append)      # the handle stays in fp32: one run, silent
    assert out == "result[fp32]" and clamped == 0 and len(lines) == 1
    h = Handle(1)
    with pytest.raises(RangeError, match="--strict"):
        run_guarded("reverse loop", h.run, h.read, h.to_fp32, strict=True, log=lines.append)
    assert h.runs == ["f16x2"] and h.mode == "f16x2"

    class Broken(Handle):                    # a counter that fires in the exact mode is a library bug, not a range problem
        def run(self):
            self.runs.append(self.mode); self.pending += 1
            return None
    b = Broken(1)
    with pytest.raises(RuntimeError, match="no range limit"):
        run_guarded("stage", b.run, b.read, b.to_fp32, log=lines.append)


def test_drivers_expose_the_strict_flag():
    """Both front ends (examples/generate.py and the reference-flag command lines behind sample/_common.py) accept --strict."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("generate_for_flags", os.path.join(root, "examples", "generate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.parse(["uncond", "--strict"]).strict is True and mod.parse(["uncond"]).strict is False
    from sample import _common
    assert _common.generate_args(["--model_path", "m.pt", "--strict"]).strict is True
    assert _common.generate_args(["--model_path", "m.pt"]).strict is False

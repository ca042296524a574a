"""GPU parity tests (run with -m gpu on an MI355X): fused HIP decoder / gradient / grid filler
against the CPU oracle and the reference-made golden vectors, through the C ABI."""
import hashlib

import os

import numpy as np
import pytest
import torch
from conftest import spawn_bounded

from oracle import decoder as odec
from oracle import gridfiller as ogrid
from surfd_amd import synth
from surfd_amd.spec import DecoderConfig

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _decoder(D):
    from surfd_amd.cbndec import CbnDecoder
    dec = CbnDecoder(63, D, 512, 5)
    sd = synth.synth_decoder_state_dict(DecoderConfig(latent_dim=D))
    dec.load_state_dict(sd, strict=True)
    return dec.cuda().eval(), sd


def _cos(a, b):
    return (a * b).sum(-1)


def _check_directions(c, label=""):
    """Stated tolerance on -normalize(grad udf): cosine >= 1 - 1e-5, asserted for >= 99.9 % of the
    points — the measured level: the field is piecewise linear in 11 x 512 ReLU units, and a point whose
    pre-activation is within fp32 rounding of a kink takes the other branch in a different-but-equally-valid fp32
    evaluation order.  Measured on the reference itself (tools/make_golden.py g8flips -> tests/golden/g8_direction_flips.json):
    its own fp32 sample_grads against an fp64 copy of the same decoder disagrees (cos <= 1 - 1e-5) on 0.024 % of the G8
    points at D=32 (worst cosine 0.9935) and on none at D=64; this library against the reference's fp32 values: 0.06 %
    observed.  The asserted bound is 0.1 % (round 4: 0.2 %) — 4x the reference's own flip rate, under 2x the level observed
    here — and the worst cosine must stay above 0.99 (the reference's own worst: 0.9935; round 4 asserted 0.9).  Fraction and
    worst cosine are printed with every run."""
    c = np.asarray(c)
    if c.size == 0:
        return
    bad = int((c <= 1 - 1e-5).sum())
    print(f"directions{label}: n={c.size} cos>1-1e-5 on {100.0 * (1 - bad / c.size):.3f} % of points, "
          f"median 1-{1 - np.median(c):.1e}, worst cosine {c.min():.6f}")
    # 0.1 % of the points; samples of a few thousand points get the 3-sigma counting allowance of that rate on top (4 flips in
    # 3 000 points have been seen: 0.13 % of a sample whose expectation at 0.05 % is 1.5)
    allowed = np.ceil(0.001 * c.size) if c.size >= 20000 else np.ceil(0.001 * c.size + 3.0 * np.sqrt(0.001 * c.size))
    assert bad <= max(1, int(allowed)), (bad, c.size)
    assert np.median(c) > 1 - 1e-6
    assert c.min() > 0.99


def test_library_and_device():
    from surfd_amd import _native as N
    assert N.lib().surfd_abi_version() == 1
    assert N.lib().surfd_device_count() >= 1


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
@pytest.mark.parametrize("D", [32, 64])
def test_decoder_vs_golden(golden, D, precision):
    """Both forward arithmetics (include/surfd_hip.h, surfd_decoder_set_precision) against the reference's
    own outputs, same tolerances."""
    from surfd_amd.cbndec import make_udf_func
    g = golden(f"g8_decoder_D{D}")
    dec, _ = _decoder(D)
    dec.set_precision(precision)
    lat = T(g["lat"]).cuda()
    pts = T(g["pts"]).cuda()
    dec.bind_latents(lat)
    logit = dec._logits_xyz(pts, 0).cpu().numpy()
    np.testing.assert_allclose(logit, g["logit"], rtol=1e-5, atol=3e-5)
    udf = dec.udf(pts, 0).cpu().numpy()
    np.testing.assert_allclose(udf, g["udf"], rtol=0, atol=1e-6)          # stated tolerance: 1e-6 on [0, 0.1]
    udf2, ng = dec.udf_and_ngrad(pts, 0)
    np.testing.assert_array_equal(udf2.cpu().numpy(), udf)                    # forward and gradient kernel share the forward arithmetic
    ng = ng.cpu().numpy()
    nz = np.linalg.norm(g["ngrad"], axis=-1) > 0
    _check_directions(_cos(ng, g["ngrad"])[nz])
    assert np.abs(ng[~nz]).max(initial=0.0) == 0.0
    np.testing.assert_allclose(np.linalg.norm(ng[nz], axis=-1), 1.0, atol=1e-5)
    f = make_udf_func(dec, lat)
    np.testing.assert_array_equal(f(pts).cpu().numpy(), udf)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 70001])
def test_decoder_ragged_vs_oracle(n):
    dec, sd = _decoder(32)
    g = torch.Generator().manual_seed(n)
    lat = torch.randn(1, 32, generator=g) * 0.8
    pts = torch.rand(n, 3, generator=g) * 2 - 1
    dec.bind_latents(lat.cuda())
    udf, ng = dec.udf_and_ngrad(pts.cuda(), 0)
    m = min(n, 3000)
    f = odec.make_udf_func(sd, lat)
    ref = odec.sample_udf(f, pts[:m], 2 ** 16)
    np.testing.assert_allclose(udf[:m].cpu().numpy(), ref.numpy(), rtol=0, atol=1e-6)
    refg = odec.sample_grads(f, pts[:m], 2 ** 12).numpy()
    nz = np.linalg.norm(refg, axis=-1) > 0
    _check_directions(_cos(ng[:m].cpu().numpy(), refg)[nz])
    # tile-position independence: the same point gives the same bits wherever it sits
    perm = torch.randperm(n, generator=g)
    udf_f = dec.udf(pts.cuda(), 0)                       # forward kernel (same arithmetic as the gradient kernel's forward half)
    np.testing.assert_allclose(udf_f.cpu().numpy(), udf.cpu().numpy(), rtol=0, atol=2e-7)
    udf_p = dec.udf(pts[perm].cuda(), 0)
    np.testing.assert_array_equal(udf_p.cpu().numpy(), udf_f.cpu().numpy()[perm.numpy()])


def test_encoder_vs_golden_on_the_gpu(golden):
    """a13: CoordsEncoder.encode (AutoEncoder/models/coordsenc.py:25-51) on the device against the reference's own output
    (G7) — the product's dense encoder, and through it the columns the fused kernels rebuild per tile."""
    from surfd_amd.cbndec import CoordsEncoder
    g = golden("g7_encode")
    enc = CoordsEncoder()
    pts = torch.from_numpy(g["pts"]).cuda()
    out = enc.encode_dense(pts)
    assert out.is_cuda and tuple(out.shape) == tuple(g["enc"].shape)
    np.testing.assert_allclose(out.cpu().numpy(), g["enc"], rtol=0, atol=2e-6)      # sin / cos of up to 2^9 * pi * x in fp32
    lazy = enc.encode(pts)                                   # what the reference's call site gets: shape-compatible, dense on demand
    assert tuple(lazy.shape) == tuple(g["enc"].shape)
    np.testing.assert_array_equal(lazy.materialize().cpu().numpy(), out.cpu().numpy())


def test_decoder_saturation_edge_cases():
    """logit >~ 17 -> udf == 0.0 exactly and a zero gradient vector."""
    dec, sd = _decoder(32)
    sd = dict(sd)
    sd["decoder.fc_out.bias"] = torch.tensor([60.0])
    dec.load_state_dict(sd, strict=True)
    lat = torch.zeros(1, 32)
    pts = torch.rand(500, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1
    dec.bind_latents(lat.cuda())
    udf, ng = dec.udf_and_ngrad(pts.cuda(), 0)
    f = odec.make_udf_func(sd, lat)
    ref = odec.sample_udf(f, pts, 2 ** 16)
    refg = odec.sample_grads(f, pts, 2 ** 12)
    sat = ref == 0
    assert sat.any()
    assert (udf.cpu()[sat] == 0).all()
    assert (ng.cpu()[(refg.abs().sum(-1) == 0)] == 0).all()


def test_decoder_emb_and_reference_closure():
    from surfd_amd.cbndec import CoordsEncoder
    from surfd_amd.meshudf import sample_grads, sample_udf
    dec, sd = _decoder(32)
    enc = CoordsEncoder()
    assert enc.out_dim == 63
    g = torch.Generator().manual_seed(3)
    lat = (torch.randn(1, 32, generator=g) * 0.8).cuda()
    pts = (torch.rand(777, 3, generator=g) * 2 - 1).cuda()

    def udf_func(c):                       # verbatim shape of sample/generate_uncond.py:96-101
        c = enc.encode(c.unsqueeze(0))
        p = dec(c, lat).squeeze(0)
        p = torch.sigmoid(p)
        return (1 - p) * 0.1

    a = sample_udf(udf_func, pts, 256)
    dec.bind_latents(lat)
    b = dec.udf(pts, 0)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-8)
    emb = enc.encode_dense(pts)[None]
    c = dec(emb, lat)[0]
    np.testing.assert_allclose(c.cpu().numpy(), dec._logits_xyz(pts, 0).cpu().numpy(), rtol=1e-5, atol=2e-5)
    ga = sample_grads(udf_func, pts, 200)          # autograd through the HIP reverse sweep
    gb = dec.udf_and_ngrad(pts, 0)[1]
    assert _cos(ga.cpu().numpy(), gb.cpu().numpy()).min() > 1 - 1e-6


def _sha(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


@pytest.mark.parametrize("N", [64, 128, 256])
def test_grid_callback_analytic_bit_exact(golden, N):
    """Index kernels pinned independently of the decoder: an analytic field evaluated by the
    host callable; the returned grid must equal the reference's bit for bit."""
    from surfd_amd.meshudf import GridFiller
    g = golden("g10_grid_analytic")

    def field(c):
        return ogrid.analytic_field(c.cpu()).cuda()

    gf = GridFiller(N)
    udf, grads = gf.fill_grid(field, 2 ** 30)
    assert gf.last_stats["fwd_per_level"] == list(g[f"N{N}_fwd_per_level"])
    assert gf.last_stats["grad"] == int(g[f"N{N}_grad_points"])
    # bit-exact against the oracle evaluated on this host (the analytic field is computed by the host
    # CPU on both sides; across hosts torch's CPU kernels may differ in the last ulp, so the golden
    # made in the build container is compared with a 1-ulp tolerance instead of by hash)
    ref, rgrads, rstats = ogrid.fill_grid(ogrid.analytic_field, N, 2 ** 30)
    assert rstats["fwd_per_level"] == gf.last_stats["fwd_per_level"] and rstats["grad"] == gf.last_stats["grad"]
    assert torch.equal(udf.cpu(), ref)
    np.testing.assert_allclose(grads.cpu().numpy(), rgrads.numpy(), rtol=0, atol=1e-6)
    assert (grads.cpu().abs().sum(-1) > 0).eq(rgrads.abs().sum(-1) > 0).all()
    assert float(udf.double().sum()) == pytest.approx(float(g[f"N{N}_udf_sum"]), rel=1e-7)
    if N == 64:
        np.testing.assert_allclose(udf.cpu().numpy(), g["N64_udf"], rtol=0, atol=1e-7)
        np.testing.assert_allclose(grads.cpu().numpy(), g["N64_grads_f16"].astype(np.float32), atol=1e-3)


def test_grid_callback_analytic_512_counts():
    """BASELINE's full grid size.  The reference's own run of GridFiller(512) on the analytic field
     evaluates [32768, 229376, 91350, 346066, 1349754] points per level, 746008 gradient
    points, and its udf grid sums to 13 017 254.6665; the index kernels must reproduce the counts exactly."""
    from surfd_amd.meshudf import GridFiller

    def field(c):
        return ogrid.analytic_field(c.cpu()).cuda()

    gf = GridFiller(512)
    udf, grads = gf.fill_grid(field, 2 ** 30)
    assert gf.last_stats["fwd_per_level"] == [32768, 229376, 91350, 346066, 1349754]
    assert gf.last_stats["grad"] == 746008
    assert float(udf.double().sum()) == pytest.approx(13017254.6665, rel=1e-7)
    has = grads.abs().sum(-1) > 0
    assert int(has.sum()) <= 746008 and bool((udf[has] < 2.5 * 2.0 / 512).all())
    del udf, grads
    torch.cuda.empty_cache()


@pytest.mark.parametrize("D", [32, 64])
def test_grid_native_vs_golden_and_callback(golden, D):
    """GridFiller(64) with the native decoder against the grid the REFERENCE produced, for both latent widths
    (D=64 is what configs C4 / C5 decode)."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    g = golden("g9_grid64_decoder" if D == 32 else "g9_grid64_decoder_D64")
    dec, sd = _decoder(D)
    lat = T(g["lat"]).cuda()
    f = make_udf_func(dec, lat)
    gf = GridFiller(64)
    udf, grads = gf.fill_grid(f, 2 ** 16)
    st_native = dict(gf.last_stats)
    ref = g["udf"]
    close = np.isclose(udf.cpu().numpy(), ref, rtol=0, atol=1e-6)
    assert close.mean() >= 0.9999, close.mean()
    sub = g["grad_idx"]
    mine = grads.reshape(-1, 3).cpu().numpy()[sub]
    both = (np.linalg.norm(mine, axis=-1) > 0) & (np.linalg.norm(g["grad_sub"], axis=-1) > 0)
    agree = (np.linalg.norm(mine, axis=-1) > 0) == (np.linalg.norm(g["grad_sub"], axis=-1) > 0)
    assert agree.mean() >= 0.9999
    _check_directions(_cos(mine, g["grad_sub"])[both])

    # the same decoder through the host-callback path must give identical bits
    def plain(c):
        return dec.udf(c, 0)
    udf_cb, grads_cb = GridFiller(64).fill_grid(plain, 2 ** 12, with_grads=False)
    np.testing.assert_array_equal(udf_cb.cpu().numpy(), udf.cpu().numpy())
    assert st_native["fwd_per_level"][0] == 32768 and st_native["fwd_per_level"][1] == 229376


@pytest.mark.parametrize("D", [32, 64])
def test_grid_native_properties_256(D):
    """Full-size properties (no oracle can run this in seconds): per-level counts are
    self-consistent, pruned blocks are constant, every gradient is unit or zero and sits
    exactly on the voxels below the gradient threshold.  D=64: the decoder of configs C4 / C5."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    dec, sd = _decoder(D)
    lat = (torch.randn(1, D, generator=torch.Generator().manual_seed(9)) * 0.8).cuda()
    gf = GridFiller(256)
    udf, grads = gf.fill_grid(make_udf_func(dec, lat), 2 ** 16)
    st = gf.last_stats
    assert st["fwd_per_level"][0] == 32 ** 3 and st["fwd_per_level"][1] == 7 * 32 ** 3
    assert all(c % 7 == 0 for c in st["fwd_per_level"][1:])
    thr = 2.5 * 2.0 / 256
    gnorm = grads.norm(dim=-1)
    has = gnorm > 0
    assert int(has.sum()) <= st["grad"]
    assert bool((udf[has] < thr).all())
    assert torch.allclose(gnorm[has], torch.ones_like(gnorm[has]), atol=1e-5)
    # idempotence: a second fill with the same latent returns the same bits
    udf2, grads2 = gf.fill_grid(make_udf_func(dec, lat), 2 ** 16)
    assert torch.equal(udf, udf2) and torch.equal(grads, grads2)
    # re-evaluating sampled voxels directly reproduces the stored value where it was evaluated
    idx = torch.randint(0, 256 ** 3, (20000,), generator=torch.Generator().manual_seed(1))
    i, j, k = idx // (256 * 256), (idx // 256) % 256, idx % 256
    ax = ogrid.axis_coords(256)
    pts = torch.stack([ax[i], ax[j], ax[k]], 1).cuda()
    direct = dec.udf(pts, 0)
    stored = udf.reshape(-1)[idx.cuda()]
    same = direct == stored
    # voxels that differ must lie in pruned blocks: their stored value is a coarse copy >= the finest refine threshold
    assert bool((stored[~same] >= float(torch.tensor(1.5 * 1.7 * (2.0 / 128), dtype=torch.float32))).all())
    assert float(same.float().mean()) > 0.01


def test_dense_grid_variant():
    """get_udf_and_grads (use_fast_grid_filler=False): every voxel forward, gradients below max_dist-1e-3."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller, get_udf_and_grads
    dec, sd = _decoder(32)
    lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(21)) * 0.8).cuda()
    f = make_udf_func(dec, lat)
    udf, grads = get_udf_and_grads(f, (-1, 1), 0.1, 64, 2 ** 16)
    ax = ogrid.axis_coords(64)
    pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).cuda()
    direct = dec.udf(pts, 0)                        # forward kernel: what the grid holds, bit for bit
    udf_g, ng = dec.udf_and_ngrad(pts, 0)           # gradient kernel
    assert torch.equal(udf.reshape(-1), direct)
    assert float((udf_g - direct).abs().max()) < 2e-7
    thr = float(torch.tensor(0.1 - 1e-3, dtype=torch.float32))
    want = direct < thr
    # same points, different tiling (compacted list vs all voxels): f16x2 directions agree to the last bits (per-tile
    # power-of-two scaling of the adjoint), fp32 directions exactly
    assert float((grads.reshape(-1, 3)[want] * ng[want]).sum(-1).min()) >= 1 - 1e-6
    assert bool((grads.reshape(-1, 3)[~want] == 0).all())
    dec.set_precision("fp32")
    try:
        _, grads32 = get_udf_and_grads(f, (-1, 1), 0.1, 64, 2 ** 16)
        direct32 = dec.udf(pts, 0)
        _, ng32 = dec.udf_and_ngrad(pts, 0)
        assert torch.equal(grads32.reshape(-1, 3)[direct32 < thr], ng32[direct32 < thr])
    finally:
        dec.set_precision("f16x2")
    # spot check against the oracle's dense variant on a slab
    fo = odec.make_udf_func(sd, lat.cpu())
    ref = odec.sample_udf(fo, pts[:4096].cpu(), 2 ** 16)
    np.testing.assert_allclose(udf.reshape(-1)[:4096].cpu().numpy(), ref.numpy(), rtol=0, atol=1e-6)


def test_sharded_field_single_rank_callback_path():
    """parallel.ShardedField (grid-shard mode) through the device grid filler's callback path; with one
    rank it must reproduce the native fused fill: values bit for bit in both precision modes, gradients bit for bit in
    fp32 mode.  In f16x2 mode the reverse sweep scales each 64-point tile's adjoint by one power of two before the fp16
    split, so a point's direction can move in the last bits with the company it is tiled with (same ReLU gates, the
    forward half is bit-identical): asserted to cosine >= 1 - 1e-6 everywhere."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    from surfd_amd.parallel import ShardedField
    dec, sd = _decoder(32)
    lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(22)) * 0.8).cuda()
    f = make_udf_func(dec, lat)
    udf_n, grads_n = GridFiller(64).fill_grid(f, 2 ** 16)
    udf_s, grads_s = GridFiller(64).fill_grid(ShardedField(f), 2 ** 14)
    assert torch.equal(udf_n, udf_s)
    nz = grads_n.reshape(-1, 3).abs().sum(-1) > 0
    assert torch.equal(nz, grads_s.reshape(-1, 3).abs().sum(-1) > 0)
    c = (grads_n.reshape(-1, 3)[nz] * grads_s.reshape(-1, 3)[nz]).sum(-1)
    same = float((grads_n == grads_s).all(-1).float().mean())
    print(f"sharded vs fused gradients (f16x2): {100 * same:.3f} % of voxels bit-identical, worst cosine 1-{1 - float(c.min()):.1e}")
    assert float(c.min()) >= 1 - 1e-6
    dec.set_precision("fp32")
    udf_n, grads_n = GridFiller(64).fill_grid(f, 2 ** 16)
    udf_s, grads_s = GridFiller(64).fill_grid(ShardedField(f), 2 ** 14)
    assert torch.equal(udf_n, udf_s) and torch.equal(grads_n, grads_s)


def _run_driver(argv):
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("generate_driver", os.path.join(root, "examples", "generate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.main(argv)


def _read_obj(path):
    v, f = [], []
    for line in open(path):
        if line.startswith("v "):
            v.append([float(t) for t in line.split()[1:4]])
        elif line.startswith("f "):
            f.append([int(t) - 1 for t in line.split()[1:4]])
    return np.array(v, dtype=np.float64).reshape(-1, 3), np.array(f, dtype=np.int64).reshape(-1, 3)


@pytest.mark.parametrize("mode,extra", [("uncond", []), ("cat", ["--category", "3"]), ("text", ["--guidance_param", "3.0"]),
                                        ("image", ["--watertight"])])
def test_driver_caller_contract_end_to_end(tmp_path, mode, extra):
    """The caller contract of the sample scripts, end to end on the GPU: checkpoints in the reference
    layouts on disk -> create_model_and_diffusion / load_model_wo_clip / (CFG wrapper) -> p_sample_loop ->
    CbnDecoder.load_state_dict(strict) -> get_mesh_from_udf (or the watertight path) -> OBJ files."""
    latents, written = _run_driver([mode, "--synthetic", "--num_samples", "2", "--resolution", "64", "--respacing", "ddim20",
                                    "--output_dir", str(tmp_path)] + extra)
    L = 64 if mode in ("text", "image") else 32
    assert latents.shape == (2, 1, L) and torch.isfinite(latents).all()
    assert len(written) == 2
    for path, nv, nf in written:
        v, f = _read_obj(path)
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
        if nf:
            assert f.min() >= 0 and f.max() < nv and np.isfinite(v).all()


@pytest.mark.parametrize("mode,extra", [("uncond", ["--cond_mode", "no_cond", "--num_samples", "2"]), ("cat", ["--cond_mode", "category", "--category", "5", "--num_samples", "2"]),
                                        ("text", ["--cond_mode", "text", "--prompt", "a dining chair", "--num_samples", "2", "--cond_mask_prob", "0.1", "--guidance_param", "3.0"])])
def test_sample_module_command_lines_end_to_end(tmp_path, mode, extra):
    """python -m sample.generate_{uncond,cat,text}: the reference's own command lines (README.md:39-76) run on the GPU path
    (synthetic checkpoints in the reference layouts, a short respaced chain, 64^3)."""
    import importlib
    mod = importlib.import_module(f"sample.generate_{mode}")
    latents, written = mod.main(["--model_path", str(tmp_path / "unused.pt"), "--output_dir", str(tmp_path), "--ae_dir", str(tmp_path / "unused_ae.pt"),
                                 "--resolution", "64", "--synthetic", "--respacing", "ddim20"] + extra)
    assert latents.shape == (2, 1, 64 if mode == "text" else 32) and len(written) == 2
    for path, nv, nf in written:
        v, f = _read_obj(path)
        assert v.shape == (nv, 3) and f.shape == (nf, 3)
    if mode == "text":
        assert all("a-dining-chair_" in w[0] for w in written)


def test_get_mesh_from_udf_on_a_known_surface():
    """get_mesh_from_udf with an arbitrary callable (the reference contract) on the analytic thin shell: the mesh lies
    on the surface, is returned as (float32 vertices, int64 faces) on the device, and is an OPEN surface whose border
    (the rim) survives cleaning; differentiable=True returns the same positions with a graph attached."""
    from surfd_amd.meshudf import get_mesh_from_udf

    def field(c):
        return ogrid.analytic_field(c.cpu()).to(c.device) if not c.requires_grad else _analytic_torch(c)

    def _analytic_torch(c):
        x, y, z = c[:, 0], c[:, 1], c[:, 2]
        up = (torch.sqrt(x * x + y * y + z * z) - 0.6).abs()
        rho = torch.sqrt(x * x + y * y) - 0.6
        return torch.clamp(torch.where(z >= 0, up, torch.sqrt(rho * rho + z * z)), max=0.1)
    v, t = get_mesh_from_udf(_analytic_torch, coords_range=(-1, 1), max_dist=0.1, N=128, max_batch=2 ** 16, differentiable=False)
    assert v.dtype == torch.float32 and t.dtype == torch.int64 and v.is_cuda and t.is_cuda
    assert len(v) > 10000 and int(t.max()) == len(v) - 1
    assert float(_analytic_torch(v).max()) < 1.0 / 128 + 1e-3        # every vertex (after smoothing) within ~a voxel of the surface
    v2, t2 = get_mesh_from_udf(_analytic_torch, coords_range=(-1, 1), max_dist=0.1, N=128, max_batch=2 ** 16, differentiable=True)
    assert torch.equal(t, t2)
    np.testing.assert_allclose(v2.detach().cpu().numpy(), v.cpu().numpy(), atol=1e-6)


def test_decoder_f16x2_vs_fp32_kernel(golden):
    """The default split-fp16 forward kernel against the exact-fp32 kernel on the same points: the difference
    must be of the size of fp32 rounding noise (either kernel differs from an fp64 evaluation by ~1e-6)."""
    g = golden("g8_decoder_D32")
    dec, _ = _decoder(32)
    lat = T(g["lat"]).cuda()
    pts = (torch.rand(1 << 16, 3, generator=torch.Generator().manual_seed(11)) * 2 - 1).cuda()
    dec.bind_latents(lat)
    dec.set_precision("fp32")
    ref = dec._logits_xyz(pts, 0).cpu().numpy()
    dec.set_precision("f16x2")
    out = dec._logits_xyz(pts, 0).cpu().numpy()
    d = np.abs(out - ref)
    print("f16x2 vs fp32 kernel: max |dlogit| = %.3e, mean %.3e" % (d.max(), d.mean()))
    assert d.max() < 1e-5 and d.mean() < 1e-6
    # saturation instead of inf for absurd inputs (documented range limit of the mode)
    far = torch.full((64, 3), 3.0e4, device="cuda")
    assert torch.isfinite(dec._logits_xyz(far, 0)).all()


@pytest.mark.parametrize("D", [32, 64])
def test_grid_512_properties(D):
    """The configurations at their real grid size: 512^3 coarse-to-fine fill with the native decoder — D=32 is C3 (the
    metric's grid), D=64 is C4 / C5 — level counts, gradient support, unit norms, idempotence and direct re-evaluation of
    sampled voxels."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    dec, sd = _decoder(D)
    lat = (torch.randn(1, D, generator=torch.Generator().manual_seed(19)) * 0.8).cuda()
    gf = GridFiller(512)
    udf, grads = gf.fill_grid(make_udf_func(dec, lat), 2 ** 16)
    st = gf.last_stats
    assert st["levels"] == [32, 64, 128, 256, 512]
    assert st["fwd_per_level"][0] == 32 ** 3 and st["fwd_per_level"][1] == 7 * 32 ** 3
    assert all(c % 7 == 0 for c in st["fwd_per_level"][1:])
    gnorm = grads.norm(dim=-1)
    has = gnorm > 0
    assert int(has.sum()) <= st["grad"] and bool((udf[has] < 2.5 * 2.0 / 512).all())
    assert torch.allclose(gnorm[has], torch.ones_like(gnorm[has]), atol=1e-5)
    checksum = float(udf.double().sum())
    udf2, grads2 = gf.fill_grid(make_udf_func(dec, lat), 2 ** 16)
    assert float(udf2.double().sum()) == checksum and torch.equal(udf, udf2) and torch.equal(grads, grads2)
    idx = torch.randint(0, 512 ** 3, (50000,), generator=torch.Generator().manual_seed(2))
    i, j, k = idx // (512 * 512), (idx // 512) % 512, idx % 512
    ax = ogrid.axis_coords(512)
    direct = dec.udf(torch.stack([ax[i], ax[j], ax[k]], 1).cuda(), 0)
    stored = udf.reshape(-1)[idx.cuda()]
    same = direct == stored
    assert bool((stored[~same] >= float(torch.tensor(1.5 * 1.7 * (2.0 / 256), dtype=torch.float32))).all())
    assert dec.saturation_count() == 0
    # ---- the ORACLE at this size: voxels the fill evaluated AT the 512 level, and voxels that carry a gradient,
    #      against the CPU restatement of the reference's decoder (meshudf.py:123-206 decides which voxels those are).
    #      A voxel with an odd index whose 256-level parent corner (all indices rounded down to even) holds a value below that
    #      level's refine threshold sits in a refined block: a parent that was itself only a copy would hold a coarser level's
    #      "not close" value (>= 0.0398), so the parent was evaluated, was close, and the voxel was evaluated at the 512 level.
    thr256 = float(torch.tensor(ogrid.refine_threshold(256), dtype=torch.float32))
    g2 = torch.Generator().manual_seed(23)
    flat = udf.reshape(-1)
    par = torch.nonzero(udf[::2, ::2, ::2].reshape(-1) < thr256).flatten()            # 256-level lattice points whose block was refined
    assert len(par) >= 64, len(par)
    pick = par[torch.randint(0, len(par), (4096,), generator=g2).cuda()]
    child = torch.randint(1, 8, (4096,), generator=g2).cuda()                         # one of the 7 voxels of the block that are new at the 512 level
    pi, pj, pk = pick // (256 * 256), (pick // 256) % 256, pick % 256
    at512 = (((2 * pi + ((child >> 2) & 1)) * 512 + (2 * pj + ((child >> 1) & 1))) * 512 + (2 * pk + (child & 1))).cpu()
    gflat = grads.reshape(-1, 3)
    gcand = torch.nonzero(has.reshape(-1)).flatten()
    gsel = gcand[torch.randperm(len(gcand), generator=g2)[:4096].cuda()].cpu()
    f = odec.make_udf_func(sd, lat.cpu())
    for name, sel in (("evaluated at the 512 level", at512), ("gradient voxels", gsel)):
        i, j, k = sel // (512 * 512), (sel // 512) % 512, sel % 512
        pts = torch.stack([ax[i], ax[j], ax[k]], 1)
        ref = odec.sample_udf(f, pts, 2 ** 16)
        got = flat[sel.cuda()].cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=1e-6, err_msg=f"D={D}: {name}")
    pts = torch.stack([ax[gsel // (512 * 512)], ax[(gsel // 512) % 512], ax[gsel % 512]], 1)
    refg = odec.sample_grads(f, pts, 2 ** 12).numpy()
    nz = np.linalg.norm(refg, axis=-1) > 0
    _check_directions(_cos(gflat[gsel.cuda()].cpu().numpy(), refg)[nz], label=f" (512^3 grid, D={D})")
    del udf, grads, udf2, grads2
    torch.cuda.empty_cache()


def test_decoder_saturation_is_counted():
    """f16x2 clamps activations at 65504; that must never be silent: the counter reports it, fp32 mode has no limit."""
    dec, _ = _decoder(32)
    pts = (torch.rand(4096, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    dec.bind_latents((torch.randn(1, 32, generator=torch.Generator().manual_seed(4)) * 0.8).cuda())
    dec.udf(pts, 0)
    assert dec.saturation_count() == 0
    dec.bind_latents(torch.full((1, 32), 200.0).cuda())      # conditional-BN scales ~50 per layer -> activations beyond fp16, inside fp32
    dec.udf(pts, 0)
    assert dec.saturation_count() > 0
    dec.set_precision("fp32")
    try:
        u = dec.udf(pts, 0)
        assert torch.isfinite(u).all() and dec.saturation_count() == 0
    finally:
        dec.set_precision("f16x2")


def test_range_guard_reruns_a_clamping_stage_in_fp32(capsys):
    """What the product drivers do with the saturation counters (surfd_amd/rangeguard.py, examples/generate.py): the lat = 200
    decoder of the test above clamps in the f16x2 mode -> the guard re-runs the stage in exact fp32, says so once, and the
    result is the fp32 kernel's; a clean stage is left alone; --strict raises.  Same for the denoiser's convs."""
    from surfd_amd.rangeguard import RangeError, run_guarded
    dec, _ = _decoder(32)
    pts = (torch.rand(4096, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    try:
        dec.bind_latents((torch.randn(1, 32, generator=torch.Generator().manual_seed(4)) * 0.8).cuda())
        u0, clamped = run_guarded("shape 0 (decoder grids)", lambda: dec.udf(pts, 0), dec.saturation_count, lambda: dec.set_precision("fp32"))
        assert clamped == 0 and capsys.readouterr().err == ""
        dec.bind_latents(torch.full((1, 32), 200.0).cuda())
        with pytest.raises(RangeError):
            run_guarded("shape 1 (decoder grids)", lambda: dec.udf(pts, 0), dec.saturation_count, lambda: dec.set_precision("fp32"), strict=True)
        u1, clamped = run_guarded("shape 1 (decoder grids)", lambda: dec.udf(pts, 0), dec.saturation_count, lambda: dec.set_precision("fp32"))
        err = capsys.readouterr().err
        assert clamped > 0 and err.count("\n") == 1 and "shape 1 (decoder grids)" in err and "exact fp32" in err
        assert torch.isfinite(u1).all() and torch.equal(u1, dec.udf(pts, 0)) and dec.saturation_count() == 0      # the handle is in fp32 now
    finally:
        dec.set_precision("f16x2")


def test_callback_point_lists_are_in_voxel_order():
    """the device lists are appended with per-wave atomics (scheduling-dependent order); what the
    callback path hands to the host — and parallel.ShardedField splits over ranks by position — must be a
    deterministic order: ascending voxel index for the gradient points, ascending parent corner for level points."""
    from surfd_amd.meshudf import GridFiller
    N = 128
    seen = []

    def field(c):
        seen.append(c.clone())
        return ogrid.analytic_field(c.cpu()).cuda()

    class WithGrads:
        def __call__(self, c):
            return field(c)

        def grads(self, c, max_batch):
            seen.append(c.clone())
            return torch.zeros(c.shape[0], 3, device=c.device)

    GridFiller(N).fill_grid(WithGrads(), 2 ** 30)
    vox = 2.0 / (N - 1)

    def flat(c):
        ijk = torch.round((c.double() + 1.0) / vox).long()
        return (ijk[:, 0] * N + ijk[:, 1]) * N + ijk[:, 2]
    gradpts = seen[-1]
    f = flat(gradpts)
    assert bool((f[1:] > f[:-1]).all()), "gradient points are not in ascending voxel order"
    for lvl_pts in seen[1:-1]:                          # levels >= 1: 7 children per parent, parents ascending
        corners = flat(lvl_pts.reshape(-1, 7, 3)[:, 0, :])
        assert bool((corners[1:] > corners[:-1]).all())


def test_mesh_512_thin_shell_end_to_end(golden):
    """SURVEY.md §8c G11 at full size: GridFiller(512) on the GPU (analytic thin-shell field through the callback
    path, bit-exact grid) -> native marching cubes on the host must give the reference's 222 793 vertices /
    444 357 faces, with the very same arrays (SHA-256 of the reference's output)."""
    import hashlib
    from surfd_amd import mcubes
    from surfd_amd.meshudf import GridFiller
    g = golden("g13_marching_cubes")

    class Field:
        def __call__(self, c):
            return ogrid.analytic_field(c.cpu()).cuda()

        def grads(self, c, max_batch):
            # the oracle's analytic gradient: -normalize(d u / d p) by autograd on the CPU restatement
            p = c.detach().cpu().clone().requires_grad_(True)
            ogrid.analytic_field(p).sum().backward()
            return (-torch.nn.functional.normalize(p.grad, dim=1)).cuda()
    udf, grads = GridFiller(512).fill_grid(Field(), 2 ** 30)
    udf[udf < 0] = 0
    import time
    t0 = time.time()
    v, f, _, _ = mcubes.udf_mc_lewiner(udf.cpu().numpy(), grads.cpu().numpy())
    print(f"native marching cubes at 512^3: {len(v)} vertices / {len(f)} faces in {time.time() - t0:.2f} s (one host core)")
    assert (len(v), len(f)) == (222793, 444357) == (int(g["thin_shell_512_nv"]), int(g["thin_shell_512_nf"]))
    # The arrays themselves: against the reference's own extension on THIS grid (oracle/_ref travels with the snapshot).
    # (torch's CPU sqrt goes through MKL VML, whose last bit differs between the CPU that made the fixture and this host's:
    # ~9 % of the near-surface udf values move by one ulp, so the fixture's vertex hash is only comparable on the same CPU.)
    from oracle import build_ref
    cy = build_ref.load()
    if cy is not None:
        rv, rf, _, _ = build_ref.reference_udf_mc(cy, mcubes.lut_tables(), udf.cpu().numpy(), grads.cpu().numpy())
        np.testing.assert_array_equal(f, rf)
        np.testing.assert_array_equal(v, rv)
    else:
        assert hashlib.sha256(np.ascontiguousarray(f, np.int32).tobytes()).hexdigest() == str(g["thin_shell_512_faces_sha256"])
    del udf, grads
    torch.cuda.empty_cache()


def test_batched_grid_fill_matches_per_shape_fill():
    """meshudf.fill_grids (one persistent decoder launch per level for ALL shapes of a batch) against GridFiller.fill_grid
    shape after shape: same bits, values and gradients, incl. a shape without gradients and both precisions."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller, fill_grids
    dec, sd = _decoder(32)
    S = 5
    lat = (torch.randn(S, 32, generator=torch.Generator().manual_seed(31)) * 0.8).cuda()
    for precision in ("f16x2", "fp32"):
        dec.set_precision(precision)
        dec.bind_latents(lat)
        ref = [GridFiller(128).fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16, with_grads=(k != 2)) for k in range(S)]
        ref = [(u.clone(), None if g is None else g.clone()) for u, g in ref]
        fillers = [GridFiller(128) for _ in range(S)]
        outs = [(torch.empty(128, 128, 128, device="cuda"), None if k == 2 else torch.empty(128, 128, 128, 3, device="cuda")) for k in range(S)]
        fill_grids(fillers, dec, list(range(S)), outs)
        for k in range(S):
            assert torch.equal(outs[k][0], ref[k][0]), (precision, k)
            if ref[k][1] is not None:
                assert torch.equal(outs[k][1], ref[k][1]), (precision, k)
            assert fillers[k]._stats()["fwd_per_level"][0] == 32 ** 3
    dec.set_precision("f16x2")
    with pytest.raises(RuntimeError):
        fill_grids([fillers[0], fillers[0]], dec, [0, 1], outs[:2])          # one handle per shape


@pytest.mark.parametrize("N", [128, 512])
def test_band_mesher_pipeline_equals_dense_marching_cubes(N):
    """f1's sparse hand-off end to end (bench.py --endpoint e2): device-side compaction of the near-surface band, copy
    over a side stream into pinned memory, host mesher threads — against get_mesh_from_udf's marching-cubes stage on the
    dense grids copied whole (meshudf.py:338-349): faces and float32 vertices bit for bit, for several shapes in flight
    at once (decoder grids at 128^3, the analytic thin shell at 512^3), and the D2H is the band, not 16 N^3 bytes."""
    from surfd_amd import mcubes
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    grids = []
    if N == 128:
        dec, sd = _decoder(32)
        lat = (torch.randn(3, 32, generator=torch.Generator().manual_seed(77)) * 0.8).cuda()
        dec.bind_latents(lat)
        for k in range(3):
            grids.append(GridFiller(N).fill_grid(make_udf_func(dec, lat[k], sample=k), 2 ** 16))
    else:
        class Field:
            def __call__(self, c):
                return ogrid.analytic_field(c.cpu()).cuda()

            def grads(self, c, max_batch):
                p = c.detach().cpu().clone().requires_grad_(True)
                ogrid.analytic_field(p).sum().backward()
                return (-torch.nn.functional.normalize(p.grad, dim=1)).cuda()
        grids.append(GridFiller(N).fill_grid(Field(), 2 ** 30))
    bm = mcubes.BandMesher(N, threads=2, slots=2, keep=True)
    try:
        for rep in range(2):                                       # slots and scratch volumes are reused
            for k, (u, g) in enumerate(grids):
                bm.submit(u, g, tag=(rep, k))
        bm.drain()
        st = bm.stats()
        assert st["meshed_shapes"] == 2 * len(grids) and st["d2h_bytes_per_shape"] < 0.1 * st["dense_d2h_bytes_per_shape"]
        got = {tag: (v, f) for tag, v, f in bm.meshes}
        for k, (u, g) in enumerate(grids):
            uc = u.clone()
            uc[uc < 0] = 0
            v0, f0, _, _ = mcubes.udf_mc_lewiner(uc.cpu().numpy(), g.cpu().numpy())
            for rep in range(2):
                v, f = got[(rep, k)]
                np.testing.assert_array_equal(f, f0)
                np.testing.assert_array_equal(v, v0.astype(np.float32))
        print(f"band mesher at {N}^3: {st['band_voxels_per_shape']:.0f} band voxels / shape, {st['d2h_bytes_per_shape'] / 1e6:.2f} MB D2H "
              f"instead of {st['dense_d2h_bytes_per_shape'] / 1e6:.0f} MB, mesh {st['mesh_s_per_shape_one_thread']:.3f} s / shape on one thread")
    finally:
        bm.close()
    # a band larger than the handle's capacity is refused loudly, not truncated
    small = mcubes.BandMesher(N, threads=1, slots=1, capacity=16)
    try:
        small.submit(*grids[0])
        with pytest.raises(RuntimeError, match="band holds"):
            small.drain()
    finally:
        small.close()
    del grids
    torch.cuda.empty_cache()


def _two_gpu_shard_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)          # RCCL over xGMI
    try:
        from surfd_amd.cbndec import make_udf_func
        from surfd_amd.meshudf import GridFiller
        from surfd_amd.parallel import ShardedField
        dec, _ = _decoder(32)
        dec.set_precision("fp32")                 # bitwise contract incl. gradients (tile company does not matter in this mode)
        lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(22)) * 0.8).cuda()
        f = make_udf_func(dec, lat)
        udf_1, grads_1 = GridFiller(64).fill_grid(f, 2 ** 16)                             # this rank alone, fused fill
        udf_s, grads_s = GridFiller(64).fill_grid(ShardedField(f), 2 ** 14)              # levels split over the ranks
        out[rank] = (torch.equal(udf_1, udf_s), torch.equal(grads_1, grads_s), udf_s.cpu(), int((udf_s < 0.05).sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="grid-shard over RCCL needs two GPUs (1-GPU boxes skip; the gloo test covers the logic)")
def test_sharded_field_two_gpus_native_decoder():
    """§8e grid-shard mode on hardware: two ranks, each evaluating half of every level's points with the native decoder,
    ncclAllGather of the values — every rank ends with the grid (and gradients) of the single-GPU fused fill, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.get_context("spawn").Manager().dict()
    spawn_bounded(_two_gpu_shard_worker, (2, port, out), 2)
    assert out[0][0] and out[0][1] and out[1][0] and out[1][1]
    assert torch.equal(out[0][2], out[1][2]) and out[0][3] == out[1][3] > 0


# ---- grid-shard mode, native (surfd_grid_shard_*: no host read between levels, voxel-ordered lists, tiles r, r + G, ...) ----
@pytest.mark.parametrize("world", [1, 2, 3])
@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_native_sharded_fill_equals_fused_fill(world, precision):
    """§8e: the native sharded fill with `world` ranks — played in turn by this process (simulate_ranks), so the tile split, the
    fixed-capacity buffers, the ordered classification and the device-side counts are all exercised on one GPU — gives the grid
    AND the gradients of the fused single-rank fill, bit for bit (same voxel-ordered gradient list, same 64-point tiles)."""
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    dec, _ = _decoder(32)
    dec.set_precision(precision)
    try:
        lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(23)) * 0.8).cuda()
        f = make_udf_func(dec, lat)
        ref = GridFiller(128)
        udf_1, grads_1 = ref.fill_grid(f, 2 ** 16)
        gf = GridFiller(128)
        udf_s, grads_s = gf.fill_grid_sharded(f, world=world, simulate_ranks=True, capacity=1 << 21)
        assert gf.last_stats == ref.last_stats and gf.last_stats["grad"] > 0
        assert torch.equal(udf_1, udf_s) and torch.equal(grads_1, grads_s)
        # a second shape on the same handle (buffers, flags and counters are reused)
        lat2 = (torch.randn(1, 32, generator=torch.Generator().manual_seed(24)) * 0.8).cuda()
        f2 = make_udf_func(dec, lat2)
        a, b = ref.fill_grid(f2, 2 ** 16)
        c, d = gf.fill_grid_sharded(f2, world=world, simulate_ranks=True, capacity=1 << 21)
        assert torch.equal(a, c) and torch.equal(b, d)
        # a level longer than the exchange buffers is reported, not silently cut
        with pytest.raises(RuntimeError, match="capacity"):
            GridFiller(128).fill_grid_sharded(f, world=world, simulate_ranks=True, capacity=4096)
        with pytest.raises(RuntimeError, match="capacity"):
            GridFiller(128).fill_grid_sharded(f, world=world, simulate_ranks=True, grad_capacity=64)
        # ... and COUNTED on the device when nobody reads the counts back (stats=False: the timed loops of bench.py): per-level
        # capacities, the last level one tile short of what this field needs
        counts = ref.last_stats["fwd_per_level"]
        gc = GridFiller(128)
        assert gc.fill_grid_sharded(f2, world=world, simulate_ranks=True, stats=False, capacity=[counts[0], 1 << 21, 1 << 21]) is not None
        gc.fill_grid_sharded(f, world=world, simulate_ranks=True, stats=False, capacity=1 << 21)
        assert gc.shard_overflows(reset=False) == 0
        st2 = gc._stats()["fwd_per_level"]
        unit = 64 * world                                          # capacities are whole 64-point tiles of every rank (rounded UP by the filler)
        gc.fill_grid_sharded(f, world=world, simulate_ranks=True, stats=False, capacity=[st2[0], st2[1], (st2[2] // unit - 1) * unit])
        assert gc.shard_overflows() == 1 and gc.shard_overflows() == 0                   # read + reset
        # adaptive capacities: every fill plans the next one's buffers from its own counts (a thin level travels as what it
        # holds, not as 2^24 points); a plan that turns out too small is noticed and the fill repeated — same bits either way
        ga = GridFiller(128)
        u1, g1 = ga.fill_grid_sharded(f, world=world, simulate_ranks=True, adaptive=True)
        caps, gcap = ga._shard_plan
        assert torch.equal(u1, udf_1) and torch.equal(g1, grads_1)
        assert all(cnt <= cap <= max(1 << 16, 2 * cnt + 64 * world) for cnt, cap in zip(ga.last_stats["fwd_per_level"][1:], caps[1:]))
        u2, g2 = ga.fill_grid_sharded(f2, world=world, simulate_ranks=True, adaptive=True)
        assert torch.equal(u2, a) and torch.equal(g2, b)
        ga._shard_plan = ([caps[0], 1 << 16, 1 << 16], 1 << 16)                           # far too small for level 2 of this field
        before = ga.shard_bytes_exchanged
        u3, g3 = ga.fill_grid_sharded(f, world=world, simulate_ranks=True, adaptive=True)
        assert torch.equal(u3, udf_1) and torch.equal(g3, grads_1) and ga.shard_overflows() == 0
        assert (ga.shard_bytes_exchanged > before) == (world > 1)                          # bytes are counted where an exchange happens
    finally:
        dec.set_precision("f16x2")


def test_grid_shard_protocol_state_is_checked():
    """the surfd_grid_shard_* calls of one fill come in a fixed order; a call that names another level than the open
    fill is at, a commit without an evaluation, a commit with another capacity than its evaluation, or anything before
    shard_begin returns SURFD_ERR_STATE instead of reading stale parents / flags / counters."""
    import ctypes as C
    from surfd_amd import _native as N
    from surfd_amd.cbndec import make_udf_func
    from surfd_amd.meshudf import GridFiller
    dec, _ = _decoder(32)
    lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(23)) * 0.8).cuda()
    f = make_udf_func(dec, lat)
    gf = GridFiller(64)
    gf.fill_grid(f, 2 ** 16)                                # allocates the handle's device state
    L, h = gf._native()
    _, dh = dec._native()
    smp = dec._bind_single(lat)
    st = N.stream()
    udf = torch.empty(64, 64, 64, device="cuda"); grads = torch.empty(64, 64, 64, 3, device="cuda")
    buf = torch.zeros(3 << 18, device="cuda")
    ERR_STATE = -2
    cap, gcap = 32 ** 3, 1 << 18                            # gradient capacity = every voxel of the 64^3 grid
    assert L.surfd_grid_shard_level_eval(h, dh, smp, 0, 0, 1, N.ptr(buf), cap, st) == ERR_STATE          # nothing open
    N.check(L.surfd_grid_shard_begin(h, N.ptr(udf), N.ptr(grads), st))
    assert L.surfd_grid_shard_level_commit(h, 0, N.ptr(buf), cap, 1, st) == ERR_STATE                       # commit before eval
    assert L.surfd_grid_shard_level_eval(h, dh, smp, 1, 0, 1, N.ptr(buf), 1 << 18, st) == ERR_STATE     # level 1 before level 0
    assert L.surfd_grid_shard_grad_eval(h, dh, smp, 0, 1, N.ptr(buf), gcap, st) == ERR_STATE            # gradients before the levels
    seg = torch.empty(3 << 18, device="cuda")
    assert L.surfd_grid_shard_pack(h, 0, 0, 2, N.ptr(buf), cap, N.ptr(seg), st) == ERR_STATE               # pack before the evaluation
    N.check(L.surfd_grid_shard_level_eval(h, dh, smp, 0, 0, 1, N.ptr(buf), cap, st))
    assert L.surfd_grid_shard_pack(h, 0, 0, 3, N.ptr(buf), cap, N.ptr(seg), st) == -1                      # 32^3 points are not whole tiles of 3 ranks (ERR_ARG)
    assert L.surfd_grid_shard_level_commit(h, 0, N.ptr(buf), cap, 3, st) == -1
    assert L.surfd_grid_shard_level_commit(h, 0, N.ptr(buf), cap - 64, 1, st) == ERR_STATE                  # another capacity than the eval
    N.check(L.surfd_grid_shard_level_commit(h, 0, N.ptr(buf), cap, 1, st))
    assert L.surfd_grid_shard_level_commit(h, 0, N.ptr(buf), cap, 1, st) == ERR_STATE                       # level 0 is closed
    N.check(L.surfd_grid_shard_level_eval(h, dh, smp, 1, 0, 1, N.ptr(buf), 1 << 18, st))
    N.check(L.surfd_grid_shard_level_commit(h, 1, N.ptr(buf), 1 << 18, 1, st))
    assert L.surfd_grid_shard_grad_commit(h, N.ptr(buf), gcap, 1, st) == ERR_STATE                          # gradient commit before its eval
    N.check(L.surfd_grid_shard_grad_eval(h, dh, smp, 0, 1, N.ptr(buf), gcap, st))
    N.check(L.surfd_grid_shard_grad_commit(h, N.ptr(buf), gcap, 1, st))
    assert L.surfd_grid_shard_level_eval(h, dh, smp, 0, 0, 1, N.ptr(buf), cap, st) == ERR_STATE          # the fill is closed
    # what the protocol produced is the fused fill's grid
    a, b = GridFiller(64).fill_grid(f, 2 ** 16)
    torch.cuda.synchronize()
    assert torch.equal(a, udf) and torch.equal(b, grads) and gf.shard_overflows() == 0


def _native_shard_worker(rank, world, port, out, backend):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)          # nccl = RCCL over xGMI; gloo: the ranks share GPU 0
    try:
        from surfd_amd.cbndec import make_udf_func
        from surfd_amd.meshudf import GridFiller
        dec, _ = _decoder(32)
        lat = (torch.randn(1, 32, generator=torch.Generator().manual_seed(22)) * 0.8).cuda()
        f = make_udf_func(dec, lat)
        udf_1, grads_1 = GridFiller(64).fill_grid(f, 2 ** 16)                                        # this rank alone, fused fill
        udf_s, grads_s = GridFiller(64).fill_grid_sharded(f, rank=rank, world=world, capacity=1 << 20)   # tiles rank, rank + world, ...
        out[rank] = (torch.equal(udf_1, udf_s), torch.equal(grads_1, grads_s), udf_s.cpu(), int((udf_s < 0.05).sum()))
    finally:
        dist.destroy_process_group()


def _run_native_shard(backend):
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.get_context("spawn").Manager().dict()
    spawn_bounded(_native_shard_worker, (2, port, out, backend), 2)
    assert out[0][0] and out[0][1] and out[1][0] and out[1][1]
    assert torch.equal(out[0][2], out[1][2]) and out[0][3] == out[1][3] > 0


def test_native_sharded_fill_two_ranks_over_gloo_on_one_gpu():
    """Two PROCESSES, one GPU, gloo: each rank evaluates every second tile, the value buffers are summed over the ranks
    (through the host with this backend), both end with the fused fill's grid and gradients bit for bit."""
    _run_native_shard("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="grid-shard over RCCL needs two GPUs (1-GPU boxes skip; the gloo variant above covers the logic)")
def test_native_sharded_fill_two_gpus_over_rccl():
    _run_native_shard("nccl")


@pytest.mark.parametrize("precision", ["f16x2", "fp32"])
def test_decoder_kernels_are_bit_stable_run_to_run(precision):
    """the forward kernel (8-wave in f16x2) and the forward + reverse-sweep kernel, 40 evaluations of the same 100 000
    points each (all CUs, several tiles per persistent workgroup): identical bits."""
    dec, _ = _decoder(32)
    dec.set_precision(precision)
    try:
        g = torch.Generator().manual_seed(31)
        lat = (torch.randn(1, 32, generator=g) * 0.8).cuda()
        pts = (torch.rand(100_000, 3, generator=g) * 2 - 1).cuda()
        dec.bind_latents(lat)
        u0 = dec.udf(pts, 0).clone()
        v0, n0 = (a.clone() for a in dec.udf_and_ngrad(pts, 0))
        for run in range(1, 40):
            assert torch.equal(dec.udf(pts, 0), u0), run
            v, n = dec.udf_and_ngrad(pts, 0)
            assert torch.equal(v, v0) and torch.equal(n, n0), run
        assert dec.saturation_count() == 0
    finally:
        dec.set_precision("f16x2")

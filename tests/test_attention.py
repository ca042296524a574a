"""a19 — CrossAttention over [b, n, c] tokens (reference modules/attention.py:152-193): the oracle restatement against
the fixture made by the reference module itself (CPU), and the native MFMA op against both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import attention as oatt
from surfd_amd import synth


def _cases(golden):
    g = golden("g14_cross_attention")
    names = sorted({k.split("__")[0] for k in g.files})
    for name in names:
        qd, cd, heads, dh, seed = (int(v) for v in g[name + "__cfg"])
        t = lambda key: torch.from_numpy(g[name + "__" + key]) if name + "__" + key in g.files else None
        sd = synth.synth_cross_attention_state_dict(qd, cd or qd, heads, dh, seed=seed)
        yield name, (qd, cd or None, heads, dh), sd, t("x"), t("context"), t("mask"), t("out")


def test_oracle_cross_attention_vs_reference_fixture(golden):
    n = 0
    for name, cfg, sd, x, ctx, mask, out in _cases(golden):
        y = oatt.cross_attention(sd, cfg[2], x, ctx, mask)
        assert float((y - out).abs().max()) <= 2e-6 * max(1.0, float(out.abs().max())), name
        n += 1
    assert n == 5


def test_module_refuses_cpu_tensors():
    from surfd_amd.attention import CrossAttention
    m = CrossAttention(32, 16, heads=2, dim_head=8)
    assert sorted(m.state_dict()) == ["to_k.weight", "to_out.0.bias", "to_out.0.weight", "to_q.weight", "to_v.weight"]
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 32))


@pytest.mark.gpu
def test_native_cross_attention_vs_reference_fixture(golden):
    from surfd_amd.attention import CrossAttention
    for name, (qd, cd, heads, dh), sd, x, ctx, mask, out in _cases(golden):
        m = CrossAttention(qd, cd, heads=heads, dim_head=dh)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().eval()
        y = m(x.cuda(), None if ctx is None else ctx.cuda(), None if mask is None else mask.cuda()).cpu()
        err = float((y - out).abs().max())
        assert err <= 2e-5 * max(1.0, float(out.abs().max())), (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,m,qd,cd,heads,dh", [(4, 256, 256, 224, None, 8, 28), (2, 33, 129, 64, 48, 4, 16), (1, 300, 1, 40, 24, 5, 64),
                                                     (8, 32, 77, 448, 512, 8, 56)])
def test_native_cross_attention_vs_oracle(b, n, m, qd, cd, heads, dh):
    from surfd_amd.attention import CrossAttention
    sd = synth.synth_cross_attention_state_dict(qd, cd or qd, heads, dh, seed=n)
    g = torch.Generator().manual_seed(n * 7 + m)
    x = torch.randn(b, n, qd, generator=g)
    ctx = None if cd is None else torch.randn(b, m, cd, generator=g)
    mask = torch.rand(b, m if cd is not None else n, generator=g) > 0.25
    mod = CrossAttention(qd, cd, heads=heads, dim_head=dh)
    mod.load_state_dict(sd, strict=True)
    mod = mod.cuda().eval()
    for mk in (None, mask):
        ref = oatt.cross_attention(sd, heads, x, ctx, mk)
        y = mod(x.cuda(), None if ctx is None else ctx.cuda(), None if mk is None else mk.cuda()).cpu()
        assert float((y - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    # argument checks surface as Python errors, the handle refuses what the kernel does not cover
    with pytest.raises(ValueError):
        mod(torch.zeros(b, n, qd + 1).cuda())
    with pytest.raises(RuntimeError):
        CrossAttention(64, 64, heads=2, dim_head=256).cuda()(torch.zeros(1, 4, 64).cuda())
    assert mod(torch.zeros(0, n, qd).cuda()).shape == (0, n, qd)


@pytest.mark.gpu
def test_spatial_transformer_vs_reference_fixture(golden):
    """a19 wrappers (modules/attention.py:37-64, 196-261): SpatialTransformer with BasicTransformerBlock / GEGLU around the
    native attention op reproduces the reference module's output on seeded weights (self-attention depth 1, cross-attention
    depth 2), and carries the reference's parameter names."""
    from surfd_amd.attention import SpatialTransformer
    g = golden("g17_spatial_transformer")
    for name in ("self_d1", "cross_d2"):
        c, heads, dh, depth, ctx_dim, b, h, w, m = (int(v) for v in g[name + "__cfg"])
        mod = SpatialTransformer(c, heads, dh, depth=depth, context_dim=ctx_dim or None).eval()
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        assert "transformer_blocks.0.ff.net.0.proj.weight" in shapes and "transformer_blocks.0.attn2.to_k.weight" in shapes
        mod.load_state_dict(synth.synth_like(shapes, seed=int(g[name + "__weight_seed"]), tag="g17"), strict=True)
        mod = mod.cuda()
        x = torch.from_numpy(g[name + "__x"]).cuda()
        ctx = torch.from_numpy(g[name + "__ctx"]).cuda() if name + "__ctx" in g.files else None
        y = mod(x, context=ctx).cpu().numpy()
        ref = g[name + "__y"]
        assert np.abs(y - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (name, float(np.abs(y - ref).max()))

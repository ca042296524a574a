"""CPU model of the forward decoder kernel's default arithmetic (csrc/decoder.hip, "f16x2").

Every fp32 operand is split into two fp16 terms (weights after multiplication by one power of two SC with
max|W|*SC in [256, 512)), the products xh*wh + xh*wl + xl*wh are exact in fp32 and are accumulated in fp32.
The model below reproduces that with torch on the CPU (float64 accumulation stands in for the order-dependent
fp32 accumulation, which adds the same ~1e-7 noise to either arithmetic) and pins the accuracy claim the GPU
tests then check on the real kernel: the split evaluation is as close to an fp64 evaluation of the reference
graph as the plain fp32 evaluation is.
"""
import math

import torch
import torch.nn.functional as F

from oracle import decoder as odec
from surfd_amd import synth
from surfd_amd.spec import DecoderConfig


def _split2(x):
    h = x.to(torch.float16).float()
    return h, (x - h).to(torch.float16).float()


def _weight_scale(sd):
    m = max(float(v.abs().max()) for k, v in sd.items() if k.endswith(".weight") and ".fc_" in k and "fc_out" not in k)
    return 2.0 ** (8 - math.floor(math.log2(m)))


def _forward(sd, tab, pts, lin):
    W = lambda p: sd[p + ".weight"][:, :, 0]
    h = lin(odec.encode(pts), W("decoder.fc_p"), sd["decoder.fc_p.bias"])
    for k in range(5):
        b = f"decoder.blocks.{k}"
        n = lin(F.relu(tab[2 * k, 0] * h + tab[2 * k, 1]), W(b + ".fc_0"), sd[b + ".fc_0.bias"])
        h = h + lin(F.relu(tab[2 * k + 1, 0] * n + tab[2 * k + 1, 1]), W(b + ".fc_1"), sd[b + ".fc_1.bias"])
    u = F.relu(tab[10, 0] * h + tab[10, 1]).double()
    return F.linear(u, W("decoder.fc_out").double(), sd["decoder.fc_out.bias"].double()).squeeze(-1)


def test_split_is_exact_to_22_bits():
    x = torch.randn(1 << 16) * torch.logspace(-3, 3, 1 << 16)
    h, l = _split2(x.clamp(-65504, 65504))
    ok = x.abs() > 2.0 ** -3                       # below that the low term enters fp16's subnormal range
    rel = ((x - h - l).abs() / x.abs())[ok]
    assert float(rel.max()) <= 2.0 ** -22


def test_f16x2_matches_fp64_as_well_as_fp32_does():
    torch.manual_seed(0)
    sd = synth.synth_decoder_state_dict(DecoderConfig(latent_dim=32))
    lat = torch.randn(1, 32) * 0.8
    tab = odec.cbn_tables(sd, lat)
    tab = tab[0] if tab.dim() == 4 else tab
    pts = torch.rand(4096, 3) * 2 - 1
    sc = _weight_scale(sd)
    assert 256 <= sc * max(float(v.abs().max()) for k, v in sd.items() if k.endswith(".weight") and ".fc_" in k and "fc_out" not in k) < 512

    def lin64(x, w, b):
        return F.linear(x.double(), w.double(), b.double()).float()

    def lin32(x, w, b):
        return F.linear(x, w, b)

    def lin_split(scale):
        def lin(x, w, b):
            xh, xl = _split2(x.clamp(-65504, 65504))
            wh, wl = _split2(w * scale)
            y = (xl.double() @ wh.double().T + xh.double() @ wl.double().T) + xh.double() @ wh.double().T
            return (y / scale + b.double()).float()
        return lin

    ref = _forward(sd, tab, pts, lin64)
    e32 = (_forward(sd, tab, pts, lin32) - ref).abs()
    e16 = (_forward(sd, tab, pts, lin_split(sc)) - ref).abs()
    e16_unscaled = (_forward(sd, tab, pts, lin_split(1.0)) - ref).abs()
    assert float(e16.max()) < 1e-6 and float(e16.mean()) < 2e-7
    assert float(e16.max()) <= 2.0 * float(e32.max())            # fp32-class accuracy
    assert float(e16_unscaled.max()) > float(e16.max())           # why the weights are pre-scaled
    udf = lambda z: (1 - torch.sigmoid(z)) * 0.1
    assert float((udf(_forward(sd, tab, pts, lin_split(sc))) - udf(ref)).abs().max()) < 1e-7

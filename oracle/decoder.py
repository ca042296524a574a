"""ORACLE (test infrastructure, never shipped as product): CPU fp32 restatement of the
Surf-D UDF field — positional encoding, conditional-BatchNorm residual MLP decoder, the
``udf_func`` closure and its autograd gradient.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  Pinned by tests/golden/ (tools/make_golden.py).

Two restatements live here:
  * ``decoder_forward`` — the reference's own op graph (per-point gamma/beta 1x1 convs,
    eval-mode batch_norm, autograd gradient).  This is what is timed as the CPU baseline.
  * ``cbn_tables`` / ``decoder_forward_hoisted`` / ``udf_and_grad_analytic`` — the algebra
    the HIP kernels implement (per-sample scale/shift tables, analytic reverse sweep;
    SURVEY.md Appendix B), checked here against the first.

Reference anchors (relative to /root/reference):
  CoordsEncoder.encode           AutoEncoder/models/coordsenc.py:25-51
  ConditionalBatchNorm1d         AutoEncoder/models/cbndec.py:50-82
  ConditionalResnetBlock1d       AutoEncoder/models/cbndec.py:85-103
  DecoderConditionalBatchNorm    AutoEncoder/models/cbndec.py:35-47
  CbnDecoder.forward             AutoEncoder/models/cbndec.py:127-134
  udf_func closure               sample/generate_uncond.py:96-101
  sample_udf / sample_grads      meshudf/meshudf.py:209-251
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
UDF_MAX_DIST = 0.1
BN_EPS = 1e-5


def encode(p: Tensor, num_freqs: int = 10, max_freq_log2: int = 9) -> Tensor:
    """[..., 3] -> [..., 63]: raw xyz, then per frequency 2^j: sin(3), cos(3)."""
    freqs = 2.0 ** torch.linspace(0.0, max_freq_log2, steps=num_freqs)
    parts = [p]
    for f in freqs:
        parts += [torch.sin(p * f), torch.cos(p * f)]
    return torch.cat(parts, -1)


def _cbn(sd: SD, p: str, x: Tensor, c: Tensor) -> Tensor:
    gamma = F.conv1d(c, sd[p + ".conv_gamma.weight"], sd[p + ".conv_gamma.bias"])
    beta = F.conv1d(c, sd[p + ".conv_beta.weight"], sd[p + ".conv_beta.bias"])
    net = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], None, None, False, 0.1, BN_EPS)
    return gamma * net + beta


def _fc(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.conv1d(x, sd[p + ".weight"], sd[p + ".bias"])


def _num_blocks(sd: SD) -> int:
    k = 0
    while f"decoder.blocks.{k}.fc_0.weight" in sd:
        k += 1
    return k


def decoder_forward(sd: SD, coords_emb: Tensor, latent: Tensor) -> Tensor:
    """coords_emb [B,n,63], latent [B,D] or [B,n,D] -> logits [B,n]."""
    if latent.dim() == 2:
        latent = latent[:, None, :].expand(-1, coords_emb.shape[1], -1)
    p, c = coords_emb.transpose(1, 2), latent.transpose(1, 2)
    net = _fc(sd, "decoder.fc_p", p)
    for k in range(_num_blocks(sd)):
        b = f"decoder.blocks.{k}"
        h = _fc(sd, b + ".fc_0", F.relu(_cbn(sd, b + ".bn_0", net, c)))
        net = net + _fc(sd, b + ".fc_1", F.relu(_cbn(sd, b + ".bn_1", h, c)))
    out = _fc(sd, "decoder.fc_out", F.relu(_cbn(sd, "decoder.bn", net, c)))
    return out.squeeze(1)


def make_udf_func(sd: SD, lat: Tensor) -> Callable[[Tensor], Tensor]:
    """lat [1,D] -> callable c[n,3] -> udf[n] in [0, 0.1]."""
    def udf_func(c: Tensor) -> Tensor:
        logit = decoder_forward(sd, encode(c.unsqueeze(0)), lat).squeeze(0)
        return (1 - torch.sigmoid(logit)) * UDF_MAX_DIST
    return udf_func


def sample_udf(udf_func, coords: Tensor, max_batch: int) -> Tensor:
    out = torch.zeros(coords.shape[0])
    with torch.no_grad():
        for s in range(0, coords.shape[0], max_batch):
            out[s:s + max_batch] = udf_func(coords[s:s + max_batch])
    return out


def sample_grads(udf_func, coords: Tensor, max_batch: int) -> Tensor:
    """-normalize(d udf / d p) by autograd, chunked like the reference."""
    out = torch.zeros(coords.shape[0], 3)
    for s in range(0, coords.shape[0], max_batch):
        p = coords[s:s + max_batch].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            udf_func(p).sum().backward()
        out[s:s + max_batch] = -F.normalize(p.grad, dim=1)
    return out


# --------------------------------------------------------------------------------------
# kernel-spec algebra (what the HIP path computes)
# --------------------------------------------------------------------------------------
def cbn_layer_names(sd: SD):
    names = []
    for k in range(_num_blocks(sd)):
        names += [f"decoder.blocks.{k}.bn_0", f"decoder.blocks.{k}.bn_1"]
    return names + ["decoder.bn"]


def cbn_tables(sd: SD, lat: Tensor) -> Tensor:
    """lat [S,D] -> [S, n_cbn, 2, H]: CBN(x) == a*x + b with a = gamma/sqrt(var+eps),
    b = beta - a*mean (gamma/beta are per-sample constants because one latent is
    broadcast to every point, cbndec.py:131-132)."""
    rows = []
    for p in cbn_layer_names(sd):
        gamma = F.linear(lat, sd[p + ".conv_gamma.weight"][:, :, 0], sd[p + ".conv_gamma.bias"])
        beta = F.linear(lat, sd[p + ".conv_beta.weight"][:, :, 0], sd[p + ".conv_beta.bias"])
        a = gamma / torch.sqrt(sd[p + ".bn.running_var"] + BN_EPS)
        rows.append(torch.stack([a, beta - a * sd[p + ".bn.running_mean"]], 1))
    return torch.stack(rows, 1)


def decoder_forward_hoisted(sd: SD, tab: Tensor, pts: Tensor, keep_masks: bool = False):
    """tab [n_cbn,2,H] of one sample, pts [n,3] -> logits [n] (and ReLU masks)."""
    W = lambda p: sd[p + ".weight"][:, :, 0]
    h = F.linear(encode(pts), W("decoder.fc_p"), sd["decoder.fc_p.bias"])
    masks = []
    nb = _num_blocks(sd)
    for k in range(nb):
        b = f"decoder.blocks.{k}"
        u = tab[2 * k, 0] * h + tab[2 * k, 1]
        masks.append(u > 0)
        n = F.linear(F.relu(u), W(b + ".fc_0"), sd[b + ".fc_0.bias"])
        v = tab[2 * k + 1, 0] * n + tab[2 * k + 1, 1]
        masks.append(v > 0)
        h = h + F.linear(F.relu(v), W(b + ".fc_1"), sd[b + ".fc_1.bias"])
    u = tab[2 * nb, 0] * h + tab[2 * nb, 1]
    masks.append(u > 0)
    o = F.linear(F.relu(u), W("decoder.fc_out"), sd["decoder.fc_out.bias"]).squeeze(-1)
    return (o, masks) if keep_masks else o


def udf_and_grad_analytic(sd: SD, tab: Tensor, pts: Tensor) -> Tuple[Tensor, Tensor]:
    """Analytic reverse sweep: returns udf[n] and
    -normalize(d udf/d p)[n,3] with torch's conventions (zero vector where the fp32
    sigmoid derivative y(1-y) is exactly 0; eps 1e-12 in the normalisation)."""
    W = lambda p: sd[p + ".weight"][:, :, 0]
    o, m = decoder_forward_hoisted(sd, tab, pts, keep_masks=True)
    nb = _num_blocks(sd)
    g = W("decoder.fc_out")[0][None, :] * m[2 * nb] * tab[2 * nb, 0]
    for k in range(nb - 1, -1, -1):
        b = f"decoder.blocks.{k}"
        t = (g @ W(b + ".fc_1")) * m[2 * k + 1] * tab[2 * k + 1, 0]
        g = g + (t @ W(b + ".fc_0")) * m[2 * k] * tab[2 * k, 0]
    e = g @ W("decoder.fc_p")                       # [n,63]
    do = e[:, 0:3].clone()
    for j in range(10):
        f = float(2 ** j)
        do = do + f * (torch.cos(pts * f) * e[:, 3 + 6 * j:6 + 6 * j] - torch.sin(pts * f) * e[:, 6 + 6 * j:9 + 6 * j])
    y = torch.sigmoid(o)
    s = (-UDF_MAX_DIST) * ((1 - y) * y)
    gvec = s[:, None] * do
    ngrad = -gvec / gvec.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return (1 - y) * UDF_MAX_DIST, ngrad

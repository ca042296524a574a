"""ORACLE (test infrastructure, never shipped as product): CPU fp32 restatement of the
Surf-D latent denoiser — a functional, state_dict-driven rewrite of the reference's op
graph.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned against the reference itself: tools/make_golden.py imports /root/reference in the
build container and commits input/output vectors under tests/golden/; tests/test_oracle_golden.py
checks this file against them.

Reference anchors (paths relative to /root/reference):
  timestep_embedding        utils/ldm_utils.py:165-185
  GroupNorm32 (32 groups)   utils/ldm_utils.py:213-230
  ResBlock._forward         models/openaimodel.py:255-275
  AttentionBlock._forward   models/openaimodel.py:318-324
  QKVAttentionLegacy        models/openaimodel.py:356-372
  Downsample / Upsample     models/openaimodel.py:91-119, 134-160
  UNetModel.forward         models/openaimodel.py:710-749
  MDM.forward               models/mdm.py:91-110
  ClassifierFreeSampleModel models/cfg_sampler.py:19-26
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def timestep_embedding(t: Tensor, dim: int = 224, max_period: float = 10000.0) -> Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _conv(sd: SD, p: str, x: Tensor, stride: int = 1) -> Tensor:
    w = sd[p + ".weight"]
    return F.conv1d(x, w, sd[p + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def res_block(sd: SD, p: str, x: Tensor, emb: Tensor) -> Tensor:
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x)))
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h)))
    if p + ".skip_connection.weight" in sd:
        x = _conv(sd, p + ".skip_connection", x)
    return x + h


def attention_block(sd: SD, p: str, x: Tensor, num_heads: int = 8) -> Tensor:
    b, c, length = x.shape
    qkv = _conv(sd, p + ".qkv", _gn(sd, p + ".norm", x))
    ch = c // num_heads
    # head-major split: each head owns a contiguous [q | k | v] run of 3*ch channels
    q, k, v = qkv.reshape(b * num_heads, 3 * ch, length).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, c, length)
    return x + _conv(sd, p + ".proj_out", a)


def _has(sd: SD, prefix: str) -> bool:
    return any(k.startswith(prefix) for k in sd)


def _run_block(sd: SD, prefix: str, h: Tensor, emb: Tensor, num_heads: int) -> Tensor:
    j = 0
    while True:
        p = f"{prefix}.{j}"
        if p + ".in_layers.0.weight" in sd:
            h = res_block(sd, p, h, emb)
        elif p + ".qkv.weight" in sd:
            h = attention_block(sd, p, h, num_heads)
        elif p + ".op.weight" in sd:
            h = _conv(sd, p + ".op", h, stride=2)
        elif p + ".conv.weight" in sd:
            h = _conv(sd, p + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))
        elif p + ".weight" in sd:
            h = _conv(sd, p, h)
        else:
            return h
        j += 1


def unet_forward(sd: SD, x: Tensor, timesteps: Tensor, context: Optional[Tensor] = None,
                 y: Optional[Tensor] = None, root: str = "Unet", num_heads: int = 8) -> Tensor:
    r = root + "." if root else ""
    mc = sd[r + "time_embed.0.weight"].shape[1]
    emb = F.linear(timestep_embedding(timesteps, mc), sd[r + "time_embed.0.weight"], sd[r + "time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd[r + "time_embed.2.weight"], sd[r + "time_embed.2.bias"])
    assert (y is not None) == (r + "label_emb.weight" in sd)
    if y is not None:
        emb = emb + sd[r + "label_emb.weight"][y]
    if context is not None:
        emb = emb + F.linear(context, sd[r + "sketch_emb.weight"], sd[r + "sketch_emb.bias"])
    hs = []
    h = x.float()
    i = 0
    while _has(sd, f"{r}input_blocks.{i}."):
        h = _run_block(sd, f"{r}input_blocks.{i}", h, emb, num_heads)
        hs.append(h)
        i += 1
    h = _run_block(sd, f"{r}middle_block", h, emb, num_heads)
    o = 0
    while _has(sd, f"{r}output_blocks.{o}."):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"{r}output_blocks.{o}", h, emb, num_heads)
        o += 1
    assert not hs
    return _conv(sd, r + "out.2", F.silu(_gn(sd, r + "out.0", h)))


def mdm_forward(sd: SD, cond_mode: str, x: Tensor, timesteps: Tensor, y: Optional[dict] = None) -> Tensor:
    """MDM.forward dispatch.  'text' mode takes the already-encoded CLIP vector in
    y['context'] (the CLIP tower itself is an *input* to the path)."""
    y = y or {}
    if cond_mode == "no_cond":
        return unet_forward(sd, x, timesteps)
    if "sketch" in cond_mode or "img" in cond_mode or "text" in cond_mode:
        return unet_forward(sd, x, timesteps, context=y["context"])
    return unet_forward(sd, x, timesteps, y=y["action_text"])


def cfg_forward(sd: SD, cond_mode: str, x: Tensor, timesteps: Tensor, y: dict) -> Tensor:
    """Literal classifier-free combine; MDM.forward ignores y['uncond'] so both
    evaluations see identical inputs."""
    out = mdm_forward(sd, cond_mode, x, timesteps, y)
    y_uncond = dict(y)
    y_uncond["uncond"] = True
    out_uncond = mdm_forward(sd, cond_mode, x, timesteps, y_uncond)
    return out_uncond + (y["scale"].view(-1, 1, 1) * (out - out_uncond))

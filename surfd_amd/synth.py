"""Deterministic synthetic weights / noise for tests and benchmarks.

No pretrained checkpoints exist offline and a freshly initialised
reference model is degenerate: ``zero_module`` zeroes every ResBlock ``out_layers.3``,
every attention ``proj_out`` and the head conv (reference models/openaimodel.py:229-231,
312, 685) and the decoder's ``fc_1`` / ``conv_gamma`` / ``conv_beta`` weights are zero
(AutoEncoder/models/cbndec.py:62-66, 97).  Parity tests on such weights test nothing, so
both sides (reference-import golden generator, oracle, HIP path, bench) regenerate the
SAME non-degenerate tensors from the key name alone: a ``torch.Generator`` seeded with
``crc32(key) ^ seed`` (recipe: SURVEY.md §8d).  Only inputs/outputs are ever committed.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional

import torch

from .spec import DecoderConfig, UNetConfig, decoder_param_spec, unet_param_spec


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _normal(shape, std, key, seed, mean=0.0):
    return torch.randn(tuple(shape), generator=_gen(key, seed), dtype=torch.float32) * std + mean


def synth_unet_state_dict(cfg: UNetConfig = UNetConfig(), seed: int = 0, root: str = "Unet",
                          head_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """``head_gain`` scales the output head (``out.2``): with the default 1.0 the untrained map x_t -> x_{t-1} is
    chaotic (fp noise grows exponentially over a 1000-step chain); ``CONTRACTIVE_HEAD_GAIN`` makes
    |d x_{t-1} / d x_t| ~ coef2 < 1, so a 1000-step trajectory can be compared end to end."""
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in unet_param_spec(cfg, root):
        leaf = key.rsplit(".", 1)[-1]
        is_norm = (".in_layers.0." in key or ".out_layers.0." in key or ".norm." in key
                   or key.startswith(f"{root}.out.0."))
        if is_norm:
            sd[key] = _normal(shape, 0.05, key, seed, mean=1.0 if leaf == "weight" else 0.0)
        elif "label_emb" in key:
            sd[key] = _normal(shape, 0.1, key, seed)
        elif leaf == "bias":
            sd[key] = _normal(shape, 0.02, key, seed)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 0.5 if (".out_layers.3." in key or ".proj_out." in key) else 1.0
            sd[key] = _normal(shape, gain / fan_in ** 0.5, key, seed)
        if head_gain != 1.0 and key.startswith(f"{root}.out.2."):
            sd[key] = sd[key] * head_gain
    return sd


CONTRACTIVE_HEAD_GAIN = 0.05


def synth_decoder_state_dict(cfg: DecoderConfig = DecoderConfig(), seed: int = 0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in decoder_param_spec(cfg):
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_mean":
            sd[key] = _normal(shape, 0.1, key, seed)
        elif leaf == "running_var":
            sd[key] = torch.rand(tuple(shape), generator=_gen(key, seed), dtype=torch.float32) + 0.5
        elif ".conv_gamma." in key or ".conv_beta." in key:
            if leaf == "bias":
                sd[key] = _normal(shape, 0.05, key, seed, mean=1.0 if ".conv_gamma." in key else 0.0)
            else:
                sd[key] = _normal(shape, 0.3 / shape[1] ** 0.5, key, seed)
        elif leaf == "bias":
            sd[key] = _normal(shape, 0.02, key, seed)
        else:
            gain = 0.5 if ".fc_1." in key else 1.0
            sd[key] = _normal(shape, gain / shape[1] ** 0.5, key, seed)
    return sd


def synth_cross_attention_state_dict(query_dim: int, context_dim: int, heads: int, dim_head: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic weights in the state_dict layout of the reference's CrossAttention (modules/attention.py:161-169)."""
    inner = heads * dim_head
    shapes = {"to_q.weight": (inner, query_dim), "to_k.weight": (inner, context_dim), "to_v.weight": (inner, context_dim),
              "to_out.0.weight": (query_dim, inner), "to_out.0.bias": (query_dim,)}
    return {k: _normal(s, 0.1 if k.endswith("bias") else 1.5 / s[-1] ** 0.5, "xattn." + k, seed) for k, s in shapes.items()}


def synth_noise(num_steps: int, sample_index: int, latent_len: int, seed: int = 1234) -> torch.Tensor:
    """Noise stream of ONE sample: row 0 is x_T, row 1+k is the z drawn at loop step k.

    Seeded per *global* sample index so the result does not depend on how samples are
    sharded over ranks.  Shape ``[num_steps+1, 1, latent_len]``.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed * 1000003 + sample_index)
    return torch.randn(num_steps + 1, 1, latent_len, generator=g, dtype=torch.float32)


def synth_noise_batch(num_steps: int, first: int, count: int, latent_len: int, seed: int = 1234) -> torch.Tensor:
    """``[num_steps+1, count, 1, latent_len]`` for global samples first..first+count-1."""
    return torch.stack([synth_noise(num_steps, first + i, latent_len, seed) for i in range(count)], dim=1)


def synth_context(first: int, count: int, dim: int = 512, seed: int = 77) -> torch.Tensor:
    """Stand-in for the CLIP embedding: N(0, 0.3^2), per global sample."""
    rows = []
    for i in range(count):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed * 1000003 + first + i)
        rows.append(torch.randn(dim, generator=g, dtype=torch.float32) * 0.3)
    return torch.stack(rows, 0)


def synth_clip_state_dict(seed: int = 0, embed_dim: int = 512, image_resolution: int = 224, vision_layers: int = 12, vision_width: int = 768,
                          vision_patch_size: int = 32, context_length: int = 77, vocab_size: int = 49408, transformer_width: int = 512,
                          transformer_layers: int = 12) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for CLIP ViT-B/32 weights in OpenAI's checkpoint layout (no weights are available offline):
    per-tensor seeded normals with the standard deviations CLIP initialises with, so that activations stay O(1)."""
    sd: Dict[str, torch.Tensor] = {}

    def put(key, shape, std, mean=0.0):
        sd[key] = _normal(tuple(shape), std, "clip." + key, seed, mean=mean)

    def blocks(prefix, width, layers):
        attn_std, proj_std, fc_std = width ** -0.5, (width ** -0.5) * ((2 * layers) ** -0.5), (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}transformer.resblocks.{i}."
            put(p + "attn.in_proj_weight", (3 * width, width), attn_std); put(p + "attn.in_proj_bias", (3 * width,), 0.01)
            put(p + "attn.out_proj.weight", (width, width), proj_std); put(p + "attn.out_proj.bias", (width,), 0.01)
            put(p + "ln_1.weight", (width,), 0.05, 1.0); put(p + "ln_1.bias", (width,), 0.05)
            put(p + "ln_2.weight", (width,), 0.05, 1.0); put(p + "ln_2.bias", (width,), 0.05)
            put(p + "mlp.c_fc.weight", (4 * width, width), fc_std); put(p + "mlp.c_fc.bias", (4 * width,), 0.01)
            put(p + "mlp.c_proj.weight", (width, 4 * width), proj_std); put(p + "mlp.c_proj.bias", (width,), 0.01)

    grid = image_resolution // vision_patch_size
    put("visual.conv1.weight", (vision_width, 3, vision_patch_size, vision_patch_size), 0.02)
    put("visual.class_embedding", (vision_width,), vision_width ** -0.5)
    put("visual.positional_embedding", (grid * grid + 1, vision_width), vision_width ** -0.5)
    put("visual.ln_pre.weight", (vision_width,), 0.05, 1.0); put("visual.ln_pre.bias", (vision_width,), 0.05)
    blocks("visual.", vision_width, vision_layers)
    put("visual.ln_post.weight", (vision_width,), 0.05, 1.0); put("visual.ln_post.bias", (vision_width,), 0.05)
    put("visual.proj", (vision_width, embed_dim), vision_width ** -0.5)
    put("token_embedding.weight", (vocab_size, transformer_width), 0.02)
    put("positional_embedding", (context_length, transformer_width), 0.01)
    blocks("", transformer_width, transformer_layers)
    put("ln_final.weight", (transformer_width,), 0.05, 1.0); put("ln_final.bias", (transformer_width,), 0.05)
    put("text_projection", (transformer_width, embed_dim), transformer_width ** -0.5)
    sd["logit_scale"] = torch.tensor(2.6592)
    return sd


def synth_like(shapes: Dict[str, tuple], seed: int = 0, tag: str = "like") -> Dict[str, torch.Tensor]:
    """Seeded tensors for an arbitrary parameter table {key: shape}: normalisation scales around 1, other vectors small,
    matrices / conv kernels with fan-in scaling — one generator per key, so the result does not depend on key order."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape in shapes.items():
        shape = tuple(shape)
        if len(shape) <= 1:
            is_scale = "norm" in key and key.endswith("weight")
            out[key] = _normal(shape, 0.05, f"{tag}.{key}", seed, mean=1.0 if is_scale else 0.0)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[key] = _normal(shape, fan_in ** -0.5, f"{tag}.{key}", seed)
    return out

"""Parameter layout of the two networks on the sampling hot path.

The checkpoint layouts are part of the drop-in contract:

* diffusion checkpoint: flat ``state_dict`` of ``MDM`` — keys ``Unet.*`` in the
  registration order of reference ``models/openaimodel.py:504-692`` (UNetModel.__init__),
  ``models/mdm.py:34-57`` (the fixed UNet configuration).
* auto-encoder checkpoint: ``ckpt["decoder"]`` = state_dict of ``CbnDecoder``
  (reference ``AutoEncoder/models/cbndec.py:4-134``).

This module is pure Python (no torch import) so it can be used by the native
plan builder cross-check, the weight synthesiser and the nn.Module drop-ins.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

Shape = Tuple[int, ...]


@dataclass(frozen=True)
class UNetConfig:
    """Fixed by reference models/mdm.py:34-57."""
    in_channels: int = 1
    model_channels: int = 224
    out_channels: int = 1
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: Optional[int] = 512
    num_classes: Optional[int] = None

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.model_channels


@dataclass
class BlockSpec:
    """One TimestepEmbedSequential: ordered list of (kind, cin, cout)."""
    prefix: str
    layers: List[Tuple[str, int, int]] = field(default_factory=list)
    ds: int = 1          # downsample rate of the block's OUTPUT


def unet_blocks(cfg: UNetConfig) -> Tuple[List[BlockSpec], BlockSpec, List[BlockSpec]]:
    """Re-derives the block structure (reference openaimodel.py:516-680)."""
    mc = cfg.model_channels
    inputs: List[BlockSpec] = [BlockSpec("input_blocks.0", [("conv", cfg.in_channels, mc)], 1)]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            blk = BlockSpec(f"input_blocks.{len(inputs)}", [("res", ch, mult * mc)], ds)
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                blk.layers.append(("attn", ch, ch))
            inputs.append(blk)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            ds *= 2
            inputs.append(BlockSpec(f"input_blocks.{len(inputs)}", [("down", ch, ch)], ds))
            chans.append(ch)
    middle = BlockSpec("middle_block", [("res", ch, ch), ("attn", ch, ch), ("res", ch, ch)], ds)
    outputs: List[BlockSpec] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            blk = BlockSpec(f"output_blocks.{len(outputs)}", [("res", ch + ich, mc * mult)], ds)
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                blk.layers.append(("attn", ch, ch))
            if level and i == cfg.num_res_blocks:
                blk.layers.append(("up", ch, ch))
                ds //= 2
                blk.ds = ds
            outputs.append(blk)
    return inputs, middle, outputs


def _layer_params(prefix: str, kind: str, cin: int, cout: int, ted: int) -> List[Tuple[str, Shape]]:
    p: List[Tuple[str, Shape]] = []
    if kind == "conv":
        p += [(f"{prefix}.weight", (cout, cin, 3)), (f"{prefix}.bias", (cout,))]
    elif kind == "res":
        p += [(f"{prefix}.in_layers.0.weight", (cin,)), (f"{prefix}.in_layers.0.bias", (cin,)),
              (f"{prefix}.in_layers.2.weight", (cout, cin, 3)), (f"{prefix}.in_layers.2.bias", (cout,)),
              (f"{prefix}.emb_layers.1.weight", (cout, ted)), (f"{prefix}.emb_layers.1.bias", (cout,)),
              (f"{prefix}.out_layers.0.weight", (cout,)), (f"{prefix}.out_layers.0.bias", (cout,)),
              (f"{prefix}.out_layers.3.weight", (cout, cout, 3)), (f"{prefix}.out_layers.3.bias", (cout,))]
        if cin != cout:
            p += [(f"{prefix}.skip_connection.weight", (cout, cin, 1)),
                  (f"{prefix}.skip_connection.bias", (cout,))]
    elif kind == "attn":
        p += [(f"{prefix}.norm.weight", (cin,)), (f"{prefix}.norm.bias", (cin,)),
              (f"{prefix}.qkv.weight", (3 * cin, cin, 1)), (f"{prefix}.qkv.bias", (3 * cin,)),
              (f"{prefix}.proj_out.weight", (cin, cin, 1)), (f"{prefix}.proj_out.bias", (cin,))]
    elif kind == "down":
        p += [(f"{prefix}.op.weight", (cout, cin, 3)), (f"{prefix}.op.bias", (cout,))]
    elif kind == "up":
        p += [(f"{prefix}.conv.weight", (cout, cin, 3)), (f"{prefix}.conv.bias", (cout,))]
    else:  # pragma: no cover
        raise ValueError(kind)
    return p


def unet_param_spec(cfg: UNetConfig = UNetConfig(), root: str = "Unet") -> List[Tuple[str, Shape]]:
    """(key, shape) in reference state_dict order."""
    ted, mc = cfg.time_embed_dim, cfg.model_channels
    r = root + "." if root else ""
    out: List[Tuple[str, Shape]] = [
        (f"{r}time_embed.0.weight", (ted, mc)), (f"{r}time_embed.0.bias", (ted,)),
        (f"{r}time_embed.2.weight", (ted, ted)), (f"{r}time_embed.2.bias", (ted,)),
    ]
    if cfg.num_classes is not None:
        out.append((f"{r}label_emb.weight", (cfg.num_classes, ted)))
    if cfg.context_dim is not None:
        out += [(f"{r}sketch_emb.weight", (ted, cfg.context_dim)), (f"{r}sketch_emb.bias", (ted,))]
    inputs, middle, outputs = unet_blocks(cfg)
    for blk in inputs + [middle] + outputs:
        for j, (kind, cin, cout) in enumerate(blk.layers):
            out += _layer_params(f"{r}{blk.prefix}.{j}", kind, cin, cout, ted)
    out += [(f"{r}out.0.weight", (mc,)), (f"{r}out.0.bias", (mc,)),
            (f"{r}out.2.weight", (cfg.out_channels, mc, 3)), (f"{r}out.2.bias", (cfg.out_channels,))]
    return out


@dataclass(frozen=True)
class DecoderConfig:
    """CbnDecoder(63, D, 512, 5) — reference sample/generate_uncond.py:55-65."""
    input_dim: int = 63
    latent_dim: int = 32
    hidden_dim: int = 512
    num_hidden_layers: int = 5
    out_dim: int = 1


def _cbn_params(prefix: str, D: int, H: int) -> List[Tuple[str, Shape]]:
    return [(f"{prefix}.conv_gamma.weight", (H, D, 1)), (f"{prefix}.conv_gamma.bias", (H,)),
            (f"{prefix}.conv_beta.weight", (H, D, 1)), (f"{prefix}.conv_beta.bias", (H,)),
            (f"{prefix}.bn.running_mean", (H,)), (f"{prefix}.bn.running_var", (H,)),
            (f"{prefix}.bn.num_batches_tracked", ())]


def decoder_param_spec(cfg: DecoderConfig = DecoderConfig()) -> List[Tuple[str, Shape]]:
    """(key, shape) of ckpt["decoder"] in reference order (cbndec.py:16-30,56-58,88-93)."""
    H, D = cfg.hidden_dim, cfg.latent_dim
    out: List[Tuple[str, Shape]] = [("decoder.fc_p.weight", (H, cfg.input_dim, 1)), ("decoder.fc_p.bias", (H,))]
    for k in range(cfg.num_hidden_layers):
        b = f"decoder.blocks.{k}"
        out += _cbn_params(f"{b}.bn_0", D, H) + _cbn_params(f"{b}.bn_1", D, H)
        out += [(f"{b}.fc_0.weight", (H, H, 1)), (f"{b}.fc_0.bias", (H,)),
                (f"{b}.fc_1.weight", (H, H, 1)), (f"{b}.fc_1.bias", (H,))]
    out += _cbn_params("decoder.bn", D, H)
    out += [("decoder.fc_out.weight", (cfg.out_dim, H, 1)), ("decoder.fc_out.bias", (cfg.out_dim,))]
    return out


def numel(shape: Sequence[int]) -> int:
    n = 1
    for s in shape:
        n *= s
    return n

"""Argument surface of the reference's sample scripts (utils/parser_util.py: base, sampling, generate groups plus the dataset /
model / diffusion groups `generate_args` appends), translated to the one driver behind them (examples/generate.py)."""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

COND_MODE = {"uncond": "no_cond", "cat": "category", "text": "text", "image": "img", "sketch": "sketch"}


def generate_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    g = p.add_argument_group("base")
    g.add_argument("--num_actions", default=9, type=int)
    g.add_argument("--cuda", default=True, type=bool)
    g.add_argument("--device", default=0, type=int)
    g.add_argument("--seed", default=10, type=int)
    g.add_argument("--batch_size", default=64, type=int)
    g.add_argument("--distributed", default=False, type=bool)
    g = p.add_argument_group("sampling")
    g.add_argument("--model_path", required=True, type=str)
    g.add_argument("--output_dir", default="", type=str)
    g.add_argument("--num_samples", default=1, type=int)
    g.add_argument("--guidance_param", default=1.0, type=float)
    g.add_argument("--if_clip", action="store_true")
    g.add_argument("--clip_value", default=0.1, type=float)
    g = p.add_argument_group("generate")
    g.add_argument("--grid_size", default=128, type=int)
    g.add_argument("--category", default=0, type=int)
    g.add_argument("--sketch_path", default=None, type=str)
    g.add_argument("--image_path", default=None, type=str)
    g.add_argument("--mask_path", default=None, type=str)
    g.add_argument("--prompt", default=None, type=str)
    g.add_argument("--watertight", action="store_true")
    g.add_argument("--resolution", default=512, type=int)
    g.add_argument("--ae_dir", default=None, type=str)
    g = p.add_argument_group("dataset")
    g.add_argument("--dataset", default="deepfashion3d", choices=["deepfashion3d", "text2shape", "pix3d", "kcars", "shapenet"], type=str)
    g.add_argument("--data_dir", default="", type=str)
    g = p.add_argument_group("model")
    g.add_argument("--arch", default="OpenUNet", type=str)
    g.add_argument("--cond_mask_prob", default=0, type=float)
    g.add_argument("--unconstrained", action="store_true")
    g.add_argument("--cond_mode", default=None, type=str)
    g = p.add_argument_group("diffusion")
    g.add_argument("--noise_schedule", default="cosine", choices=["linear", "cosine"], type=str)
    g.add_argument("--diffusion_steps", default=1000, type=int)
    g.add_argument("--sigma_small", default=True, type=bool)
    g = p.add_argument_group("this implementation (no reference counterpart)")
    g.add_argument("--embedding", default=None, help="precomputed 512-d CLIP embedding(s) instead of running the towers")
    g.add_argument("--clip_path", default=None, help="CLIP ViT-B/32 weights for --prompt / --image_path / --sketch_path")
    g.add_argument("--bpe_path", default=None, help="CLIP's BPE vocabulary (text mode with --clip_path)")
    g.add_argument("--respacing", default="", help="e.g. ddim50 for a quick run")
    g.add_argument("--synthetic", action="store_true", help="synthetic checkpoints instead of --model_path / --ae_dir")
    g.add_argument("--strict", action="store_true",
                   help="raise instead of re-running a stage in exact fp32 when the split-fp16 kernels had to clamp an operand")
    a = p.parse_args(argv)
    if a.cond_mask_prob == 0:          # utils/parser_util.py:19-20: without it the guidance scale is forced back to 1
        a.guidance_param = 1
    return a


def run(mode: str, argv=None):
    a = generate_args(argv)
    if a.cond_mode is not None and a.cond_mode != COND_MODE[mode] and not (mode == "sketch" and a.cond_mode == "sketch"):
        raise SystemExit(f"sample.generate_{mode} drives a '{COND_MODE[mode]}' model, --cond_mode says '{a.cond_mode}'")
    if mode != "text":
        a.num_samples = a.num_samples if mode in ("uncond", "cat") else 1      # the image / sketch scripts sample one shape per call
    a.mode = mode
    if not a.output_dir:
        a.output_dir = os.path.join(os.path.dirname(os.path.abspath(a.model_path)), "generated")
    from examples import generate
    return generate.run(a)

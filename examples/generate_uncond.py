#!/usr/bin/env python3
"""The reference's sample/generate_uncond.py flow (main(), :21-123) on the MI355X drop-ins.

Only the imports differ from the reference script; checkpoints use the reference layouts
(diffusion: flat MDM state_dict; auto-encoder: {"decoder": ...}).  Without real checkpoints
(`--synthetic`) it first writes synthetic ones to --output_dir so the load path is exercised too.
The marching-cubes tail is SURVEY.md §8 f1 ("next"): the script stops at the device-resident
(udf, gradients) grids unless a `udf_mc_lewiner`-compatible callable is importable.

    python examples/generate_uncond.py --synthetic --num_samples 2 --resolution 64 --respacing ddim50
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from torch import Tensor

# --- the only lines that differ from the reference script -------------------------------------------
from surfd_amd.mdm import create_model_and_diffusion, load_model_wo_clip, ClassifierFreeSampleModel
from surfd_amd.cbndec import CoordsEncoder, CbnDecoder
from surfd_amd.meshudf import get_mesh_from_udf
# ---------------------------------------------------------------------------------------------------------


def generate_args():
    p = argparse.ArgumentParser()
    p.add_argument("--model_path", default="")
    p.add_argument("--ae_dir", default="")
    p.add_argument("--output_dir", default="gpurun_out/generate_uncond")
    p.add_argument("--num_samples", type=int, default=2)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--resolution", type=int, default=512)
    p.add_argument("--guidance_param", type=float, default=1.0)
    p.add_argument("--cond_mode", default="no_cond")
    p.add_argument("--arch", default="OpenUNet")
    p.add_argument("--num_actions", type=int, default=9)
    p.add_argument("--dataset", default="deepfashion3d")
    p.add_argument("--noise_schedule", default="cosine")
    p.add_argument("--sigma_small", default=True, type=bool)
    p.add_argument("--clip_value", type=float, default=1.0)
    p.add_argument("--seed", type=int, default=10)
    p.add_argument("--respacing", default="", help="'' = 1000 DDPM steps (reference default); e.g. ddim50")
    p.add_argument("--synthetic", action="store_true")
    return p.parse_args()


def main():
    args = generate_args()
    out_path = args.output_dir
    os.makedirs(out_path, exist_ok=True)
    torch.manual_seed(args.seed)
    assert args.num_samples <= args.batch_size
    args.batch_size = args.num_samples
    if args.synthetic:
        from surfd_amd import synth
        args.model_path = os.path.join(out_path, "model000000000.pt")
        args.ae_dir = os.path.join(out_path, "ae.pt")
        torch.save(synth.synth_unet_state_dict(), args.model_path)
        torch.save({"epoch": 0, "encoder": {}, "decoder": synth.synth_decoder_state_dict(), "optimizer": {}}, args.ae_dir)

    print("Creating model and diffusion...")
    model, diffusion = create_model_and_diffusion(args, args.respacing)
    print(f"Loading checkpoints from [{args.model_path}]...")
    state_dict = torch.load(args.model_path, map_location="cpu")
    load_model_wo_clip(model, state_dict)
    if args.guidance_param != 1:
        model = ClassifierFreeSampleModel(model)
    model.to("cuda")
    model.eval()

    cond = {"y": {}}
    ckpt = torch.load(args.ae_dir)
    latent_size = 32
    coords_encoder = CoordsEncoder()
    decoder = CbnDecoder(coords_encoder.out_dim, latent_size, 512, 5)
    decoder.load_state_dict(ckpt["decoder"], strict=True)
    decoder = decoder.cuda()
    decoder.eval()
    for param in decoder.parameters():
        param.requires_grad = False

    sample_fn = diffusion.p_sample_loop if not args.respacing.startswith("ddim") else diffusion.ddim_sample_loop
    sample = sample_fn(model, (args.batch_size, 1, latent_size), clip_denoised=False, model_kwargs=cond,
                       skip_timesteps=0, init_image=None, progress=True, noise=None)

    udf_max_dist = 0.1
    results = []
    for k in range(args.batch_size):
        lat = sample[k]

        def udf_func(c: Tensor) -> Tensor:          # verbatim closure of the reference script
            c = coords_encoder.encode(c.unsqueeze(0))
            p = decoder(c, lat).squeeze(0)
            p = torch.sigmoid(p)
            p = (1 - p) * udf_max_dist
            return p

        udf, grads = get_mesh_from_udf(udf_func, coords_range=(-1, 1), max_dist=udf_max_dist, N=args.resolution,
                                       max_batch=2 ** 16, differentiable=False)
        results.append((udf, grads))
        print(f"shape {k}: udf grid {tuple(udf.shape)} min {float(udf.min()):.4f} max {float(udf.max()):.4f}; "
              f"{int((grads.abs().sum(-1) > 0).sum())} voxels carry a gradient")
    print(f"done: {len(results)} shapes; grids are device-resident (marching cubes: SURVEY.md §8 f1, next)")
    return sample, results


if __name__ == "__main__":
    main()

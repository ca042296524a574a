#!/usr/bin/env python3
"""Shape generation from the command line on the MI355X path — one driver for the five conditioning modes of the
reference's sample/generate_{uncond,cat,text,image,sketch}.py:

    python examples/generate.py uncond  --model_path diff.pt --ae_dir ae.pt --num_samples 8 --resolution 512 --output_dir out/
    python examples/generate.py cat     --category 3 ...
    python examples/generate.py text    --embedding clip_text.pt --guidance_param 3.0 [--watertight] ...
    python examples/generate.py image   --embedding clip_image.pt ...
    python examples/generate.py sketch  --embedding clip_sketch.pt ...

What the reference scripts do around the hot path is kept: checkpoints in the reference layouts (a flat ``Unet.*`` dict;
``{"decoder": ...}``), latent 32 for uncond / cat / sketch and 64 for text / image, 1000 ancestral steps with
``clip_denoised=False``, classifier-free wrapper when ``--guidance_param != 1``, one OBJ per shape, small connected
components removed afterwards (uncond / cat / image / sketch: MeshLab-default Laplacian smoothing + components below 2500
faces; --watertight: below 5000 faces; text without --watertight: the get_mesh_from_udf mesh as it is, generate_text.py:159-171).
Conditioning: ``--embedding`` (a torch / numpy file of shape [512] or [num_samples, 512]) or, with ``--clip_path`` pointing at
CLIP ViT-B/32 weights (none exist offline), the towers of surfd_amd.clip_towers on ``--prompt`` / ``--image_path`` +
``--mask_path`` (mask / crop preprocessing: surfd_amd.preprocess) / ``--sketch_path``, evaluated ONCE before the loop.
The same code runs behind ``python -m sample.generate_{uncond,cat,text,image,sketch}`` with the reference's flag names.
``--synthetic`` writes synthetic checkpoints (surfd_amd.synth) first: a smoke run without any trained weights.
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from surfd_amd import meshproc, synth  # noqa: E402
from surfd_amd.rangeguard import run_guarded  # noqa: E402
from surfd_amd.cbndec import CbnDecoder, CoordsEncoder, make_udf_func  # noqa: E402
from surfd_amd.mdm import ClassifierFreeSampleModel, create_model_and_diffusion, load_model_wo_clip  # noqa: E402
from surfd_amd.meshudf import get_mesh_from_udf, get_watertight_mesh  # noqa: E402
from surfd_amd.spec import DecoderConfig, UNetConfig  # noqa: E402

MODES = {            # cond_mode of the denoiser, latent length, conditioning kind
    "uncond": ("no_cond", 32, None),
    "cat": ("category", 32, "label"),
    "sketch": ("sketch", 32, "embedding"),
    "text": ("text", 64, "embedding"),
    "image": ("img", 64, "embedding"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("mode", choices=sorted(MODES))
    ap.add_argument("--model_path", help="diffusion checkpoint (flat dict of Unet.* tensors)")
    ap.add_argument("--ae_dir", help="auto-encoder checkpoint ({'decoder': state_dict, ...})")
    ap.add_argument("--output_dir", default="generated")
    ap.add_argument("--num_samples", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--guidance_param", type=float, default=1.0)
    ap.add_argument("--category", type=int, default=0, help="cat mode: class id 0..8")
    ap.add_argument("--embedding", help="text / image / sketch modes: file with the 512-d CLIP embedding(s)")
    ap.add_argument("--prompt", default=None, help="text mode: only used to name the output files")
    ap.add_argument("--watertight", action="store_true", help="text / image modes: closed 0.01 level set instead of the open UDF mesh")
    ap.add_argument("--respacing", default="", help="e.g. ddim50 for a quick run (the reference always runs 1000 steps)")
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--synthetic", action="store_true", help="create synthetic checkpoints under --output_dir and use them")
    ap.add_argument("--clip_path", default=None, help="CLIP ViT-B/32 weights (state_dict / TorchScript archive) for --prompt / --image_path / --sketch_path")
    ap.add_argument("--bpe_path", default=None, help="text mode with --clip_path: CLIP's bpe_simple_vocab_16e6.txt.gz")
    ap.add_argument("--image_path", default=None)
    ap.add_argument("--mask_path", default=None)
    ap.add_argument("--sketch_path", default=None)
    ap.add_argument("--strict", action="store_true",
                    help="raise instead of re-running a stage in exact fp32 when the split-fp16 kernels had to clamp an operand")
    return ap.parse_args(argv)


def postprocess_open_mesh(mode: str) -> bool:
    """MeshLab smoothing + small-component removal after get_mesh_from_udf: every reference driver except generate_text.py,
    which writes the mesh as it comes (sample/generate_text.py:159-171)."""
    return mode != "text"


def load_embedding(path, count):
    obj = torch.load(path, map_location="cpu") if not path.endswith(".npy") else torch.from_numpy(np.load(path))
    emb = torch.as_tensor(obj).float().reshape(-1, 512)
    if emb.shape[0] == 1:
        emb = emb.expand(count, 512)
    if emb.shape[0] != count:
        raise SystemExit(f"--embedding holds {emb.shape[0]} vectors, --num_samples is {count}")
    return emb.contiguous()


def synthetic_checkpoints(out_dir, cond_mode, latent):
    os.makedirs(out_dir, exist_ok=True)
    model_path, ae_path = os.path.join(out_dir, "model_synthetic.pt"), os.path.join(out_dir, "ae_synthetic.pt")
    torch.save(synth.synth_unet_state_dict(UNetConfig(num_classes=9 if cond_mode == "category" else None)), model_path)
    torch.save({"epoch": 0, "decoder": synth.synth_decoder_state_dict(DecoderConfig(latent_dim=latent))}, ae_path)
    return model_path, ae_path


def conditioning_vectors(a, count):
    """[count, 512] conditioning of the text / image / sketch modes: a precomputed embedding file, the CLIP towers on the
    reference scripts' raw inputs (needs --clip_path), or — smoke runs only — seeded synthetic vectors."""
    if a.embedding:
        return load_embedding(a.embedding, count)
    raw = {"text": a.prompt, "image": a.image_path, "sketch": a.sketch_path}[a.mode]
    if a.clip_path is None:
        if raw is not None and not a.synthetic:
            raise SystemExit(f"{a.mode} mode: turning {raw!r} into the 512-d condition needs CLIP ViT-B/32 weights (--clip_path; none are "
                             "available offline) — or pass the embedding itself with --embedding")
        return synth.synth_context(0, count)
    from PIL import Image
    from surfd_amd import preprocess
    from surfd_amd.clip_towers import ClipTowers, SimpleTokenizer
    dev = "cuda" if torch.cuda.is_available() else "cpu"      # the towers are plain torch modules evaluated once per request
    towers = ClipTowers.from_file(a.clip_path).to(dev)
    if a.mode == "text":
        if not a.prompt:
            raise SystemExit("text mode with --clip_path needs --prompt")
        tok = SimpleTokenizer(a.bpe_path)
        emb = towers.encode_text(tok.tokenize([a.prompt] * count).to(dev))                # models/mdm.py:86-89, hoisted out of the loop
    elif a.mode == "image":
        if not a.image_path or not a.mask_path:
            raise SystemExit("image mode with --clip_path needs --image_path AND --mask_path (sample/generate_image.py:92-107 crops the "
                             "photo around its mask before the image tower sees it)")
        img = np.array(Image.open(a.image_path).convert("RGB"))
        mask = np.array(Image.open(a.mask_path).convert("1"))
        clean, _ = preprocess.masked_crops(img, mask, r=0.7)                              # sample/generate_image.py:92-107
        emb = towers.encode_image(preprocess.clip_image_tensor(clean, 224)[None].to(dev)).expand(count, -1)
    else:
        if not a.sketch_path:
            raise SystemExit("sketch mode with --clip_path needs --sketch_path")
        t = preprocess.sketch_clip_tensor(Image.open(a.sketch_path), 224)                 # sample/generate_sketch.py:30-37, 75-81
        emb = towers.encode_image(t[None].to(dev)).expand(count, -1)
    return emb.float().contiguous().cpu()


def main(argv=None):
    return run(parse(argv))


def run(a):
    cond_mode, latent, kind = MODES[a.mode]
    torch.manual_seed(a.seed)
    if a.synthetic:
        a.model_path, a.ae_dir = synthetic_checkpoints(a.output_dir, cond_mode, latent)
    if not a.model_path or not a.ae_dir:
        raise SystemExit("--model_path and --ae_dir are required (or --synthetic)")
    margs = types.SimpleNamespace(cond_mode=cond_mode, arch="OpenUNet", num_actions=9, dataset="deepfashion3d",
                                  noise_schedule="cosine", sigma_small=True, clip_value=0.1)
    model, diffusion = create_model_and_diffusion(margs, a.respacing)
    load_model_wo_clip(model, torch.load(a.model_path, map_location="cpu"))
    if a.guidance_param != 1:
        model = ClassifierFreeSampleModel(model)
    model.to("cuda").eval()

    y = {}
    if kind == "label":
        y["action_text"] = torch.full((a.num_samples,), a.category, dtype=torch.int64, device="cuda")
    elif kind == "embedding":
        y["context"] = conditioning_vectors(a, a.num_samples).cuda()
    if a.guidance_param != 1:
        y["scale"] = torch.full((a.num_samples,), a.guidance_param, device="cuda")

    decoder = CbnDecoder(CoordsEncoder().out_dim, latent, 512, 5)
    decoder.load_state_dict(torch.load(a.ae_dir, map_location="cpu")["decoder"], strict=True)
    decoder = decoder.cuda().eval()

    # Range guard (surfd_amd/rangeguard.py): the default split-fp16 kernels clamp at +-65504 and count it; the reference is
    # exact fp32 (models/mdm.py:46).  A stage that clamped is run again in the exact-fp32 mode on the same random numbers.
    strict = bool(getattr(a, "strict", False))
    core = model.model if isinstance(model, ClassifierFreeSampleModel) else model
    rng = (torch.get_rng_state(), torch.cuda.get_rng_state())

    def reverse_loop():
        torch.set_rng_state(rng[0]); torch.cuda.set_rng_state(rng[1])      # both attempts draw the same noise
        return diffusion.p_sample_loop(model, (a.num_samples, 1, latent), clip_denoised=False, model_kwargs={"y": y}, progress=True)   # as sample/generate_*.py

    latents, _ = run_guarded("reverse loop (denoiser)", reverse_loop, core.saturation_count, lambda: core.set_precision("fp32"), strict)
    decoder.bind_latents(latents.reshape(a.num_samples, latent))
    stem = (a.prompt or a.mode).replace(" ", "-").replace(".", "")[:100]
    written = []
    for k in range(a.num_samples):
        field = make_udf_func(decoder, latents[k], sample=k)

        def shape_mesh():
            if a.watertight:
                verts, faces = get_watertight_mesh(field, a.resolution, max_batch=2 ** 16)
                return meshproc.keep_components_with_at_least(verts, faces, 5000)
            v, t = get_mesh_from_udf(field, coords_range=(-1, 1), max_dist=0.1, N=a.resolution, max_batch=2 ** 16, differentiable=False)
            verts, faces = v.cpu().numpy(), t.cpu().numpy()
            if postprocess_open_mesh(a.mode):
                verts = meshproc.laplacian_smooth(verts, faces, steps=3)
                verts, faces = meshproc.keep_components_with_at_least(verts, faces, 2500)
            return verts, faces

        # once the decoder has been switched to fp32 it stays there for the remaining shapes of the request
        (verts, faces), _ = run_guarded(f"shape {k} (decoder grids)", shape_mesh, decoder.saturation_count,
                                        lambda: decoder.set_precision("fp32"), strict)
        path = os.path.join(a.output_dir, f"{stem}_{k}.obj")
        meshproc.write_obj(path, verts, faces)
        written.append((path, len(verts), len(faces)))
        print(f"{path}: {len(verts)} vertices, {len(faces)} faces")
    return latents, written


if __name__ == "__main__":
    main()

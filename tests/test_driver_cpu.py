"""examples/generate.py — the host-side logic of the five drivers, exercised on CPU: mode table, the
conditioning-vector loader and the synthetic checkpoints in the reference's layouts.  The GPU end-to-end runs are in
tests/test_gpu_decoder_grid.py."""
import importlib.util
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def drv():
    spec = importlib.util.spec_from_file_location("generate_under_test", os.path.join(ROOT, "examples", "generate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mode_table_matches_the_reference_scripts(drv):
    """cond_mode and latent length per driver: sample/generate_{uncond,cat,sketch}.py sample (B,1,32) latents with
    cond_mode no_cond / category / sketch, generate_{text,image}.py (B,1,64) with text / img; category takes a class id,
    the other conditional modes a 512-d CLIP vector."""
    assert drv.MODES == {"uncond": ("no_cond", 32, None), "cat": ("category", 32, "label"), "sketch": ("sketch", 32, "embedding"),
                         "text": ("text", 64, "embedding"), "image": ("img", 64, "embedding")}
    a = drv.parse(["text", "--embedding", "e.pt", "--guidance_param", "3.0", "--watertight", "--num_samples", "2"])
    assert (a.mode, a.guidance_param, a.watertight, a.num_samples, a.resolution) == ("text", 3.0, True, 2, 512)
    with pytest.raises(SystemExit):
        drv.parse(["video"])


def test_embedding_loader_shapes(drv, tmp_path):
    one = torch.arange(512, dtype=torch.float64)
    p = tmp_path / "one.pt"
    torch.save(one, p)
    e = drv.load_embedding(str(p), 3)                              # one vector serves every sample
    assert e.shape == (3, 512) and e.dtype == torch.float32 and e.is_contiguous() and torch.equal(e[2], one.float())
    q = tmp_path / "four.npy"
    np.save(q, np.random.default_rng(0).normal(size=(4, 1, 512)))
    assert drv.load_embedding(str(q), 4).shape == (4, 512)
    with pytest.raises(SystemExit):
        drv.load_embedding(str(q), 3)                              # count mismatch is an error, not a silent broadcast


def test_synthetic_checkpoints_have_the_reference_layouts(drv, tmp_path):
    """What --synthetic writes must load exactly like the reference's files: a flat dict of `Unet.*` tensors for the
    diffusion model (sample/generate_cat.py loads it with load_model_wo_clip) and {'decoder': state_dict} for the
    auto-encoder (generate_uncond.py:74-79)."""
    from surfd_amd.cbndec import CbnDecoder
    from surfd_amd.mdm import MDM
    model_path, ae_path = drv.synthetic_checkpoints(str(tmp_path), "category", 32)
    sd = torch.load(model_path, map_location="cpu")
    assert all(k.startswith("Unet.") for k in sd) and "Unet.label_emb.weight" in sd and sd["Unet.label_emb.weight"].shape[0] == 9
    m = MDM(cond_mode="category")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("clip_model.") for k in missing)
    ae = torch.load(ae_path, map_location="cpu")
    dec = CbnDecoder(63, 32, 512, 5)
    dec.load_state_dict(ae["decoder"], strict=True)


def test_text_mode_writes_the_mesh_as_it_comes(drv):
    """sample/generate_text.py:159-171 exports the get_mesh_from_udf mesh without MeshLab smoothing or component
    removal; the four other drivers post-process."""
    assert not drv.postprocess_open_mesh("text")
    assert all(drv.postprocess_open_mesh(m) for m in ("uncond", "cat", "image", "sketch"))


def test_sample_command_lines_take_the_reference_flags():
    """python -m sample.generate_* (README.md:39-76): the reference's flag names and defaults (utils/parser_util.py:40-176),
    including its quirk that the guidance scale is reset to 1 unless --cond_mask_prob is given (:19-20)."""
    from sample import _common as sc
    a = sc.generate_args(["--model_path", "pretrained_models/diffusion_uncond.pt", "--output_dir", "./outputs/uncond/", "--cond_mode", "no_cond",
                          "--ae_dir", "pretrained_models/ae_deepfashion3d.pt", "--num_samples", "10", "--resolution", "512"])
    assert (a.num_samples, a.resolution, a.cond_mode, a.seed, a.batch_size, a.guidance_param, a.noise_schedule, a.diffusion_steps) == \
        (10, 512, "no_cond", 10, 64, 1, "cosine", 1000)
    a = sc.generate_args(["--model_path", "m.pt", "--cond_mode", "text", "--ae_dir", "ae.pt", "--prompt", "a dining chair", "--watertight",
                          "--num_samples", "10", "--guidance_param", "3.0"])
    assert a.prompt == "a dining chair" and a.watertight and a.guidance_param == 1          # forced back: cond_mask_prob is 0
    a = sc.generate_args(["--model_path", "m.pt", "--guidance_param", "3.0", "--cond_mask_prob", "0.1"])
    assert a.guidance_param == 3.0
    a = sc.generate_args(["--model_path", "m.pt", "--cond_mode", "img", "--image_path", "demo_images/0049.jpg", "--mask_path", "demo_images/0049.png"])
    assert (a.image_path, a.mask_path, a.category, a.grid_size, a.clip_value) == ("demo_images/0049.jpg", "demo_images/0049.png", 0, 128, 0.1)
    with pytest.raises(SystemExit):
        sc.generate_args(["--cond_mode", "no_cond"])                                        # --model_path is required, as upstream
    import importlib
    for m in ("uncond", "cat", "text", "image", "sketch"):
        assert callable(importlib.import_module(f"sample.generate_{m}").main)
    with pytest.raises(SystemExit, match="drives a 'no_cond' model"):
        sc.run("uncond", ["--model_path", "m.pt", "--cond_mode", "text"])


@pytest.mark.parametrize("mode", ["text", "image", "sketch"])
def test_conditioning_vectors_run_the_towers_on_the_reference_inputs(drv, tmp_path, monkeypatch, mode):
    """The --clip_path branches of the driver (exercised by no test): PIL load -> the mode's
    preprocessing -> tower, with seeded CLIP weights (none exist offline).  The expected vector is computed here from the same
    building blocks, so a wrong crop / resize / missing mask shows up as a mismatch."""
    import types
    from PIL import Image
    from surfd_amd import preprocess, synth
    from surfd_amd import clip_towers as ct
    towers = ct.ClipTowers(synth.synth_clip_state_dict(seed=16))
    monkeypatch.setattr(ct.ClipTowers, "from_file", classmethod(lambda cls, path: towers))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    rng = np.random.default_rng(3)
    a = types.SimpleNamespace(mode=mode, embedding=None, clip_path="clip.pt", synthetic=False, prompt=None, image_path=None,
                              mask_path=None, sketch_path=None, bpe_path=None)
    if mode == "sketch":
        p = tmp_path / "sketch.png"
        Image.fromarray(rng.integers(0, 256, (180, 300), dtype=np.uint8)).save(p)      # 300 x 180, single channel
        a.sketch_path = str(p)
        want = towers.encode_image(preprocess.sketch_clip_tensor(Image.open(p), 224)[None])
    elif mode == "image":
        img = rng.integers(0, 256, (200, 260, 3), dtype=np.uint8)
        mask = np.zeros((200, 260), bool)
        mask[40:150, 90:210] = True
        pi, pm = tmp_path / "photo.png", tmp_path / "mask.png"
        Image.fromarray(img).save(pi)
        Image.fromarray(mask).save(pm)
        a.image_path = str(pi)
        with pytest.raises(SystemExit, match="mask_path"):
            drv.conditioning_vectors(a, 1)                                              # the photo alone is not enough
        a.mask_path = str(pm)
        clean, _ = preprocess.masked_crops(img, mask, r=0.7)
        want = towers.encode_image(preprocess.clip_image_tensor(clean, 224)[None])
    else:
        bpe = "/root/reference/CLIP/clip/bpe_simple_vocab_16e6.txt.gz"
        if not os.path.exists(bpe):
            pytest.skip("CLIP's BPE merges file is only present where the reference tree is")
        a.bpe_path = bpe
        with pytest.raises(SystemExit, match="prompt"):
            drv.conditioning_vectors(a, 1)
        a.prompt = "a dining chair"
        want = towers.encode_text(ct.SimpleTokenizer(bpe).tokenize([a.prompt]))
    got = drv.conditioning_vectors(a, 2)
    assert got.shape == (2, 512) and got.dtype == torch.float32 and got.is_contiguous() and got.device.type == "cpu"
    assert torch.equal(got[0], got[1])
    np.testing.assert_allclose(got[0].numpy(), want[0].float().numpy(), rtol=1e-5, atol=1e-6)

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

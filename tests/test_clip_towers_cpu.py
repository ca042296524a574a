"""f4: CLIP ViT-B/32 towers and tokenizer against fixtures made with the reference's vendored CLIP model class and
tokenizer (tools/make_golden.py g16; seeded weights — no CLIP checkpoint exists offline)."""
import os

import numpy as np
import pytest
import torch

from surfd_amd import synth
from surfd_amd.clip_towers import ClipTowers, SimpleTokenizer

BPE = "/root/reference/CLIP/clip/bpe_simple_vocab_16e6.txt.gz"       # CLIP's merges file is not part of this repository


@pytest.fixture(scope="module")
def towers():
    t = ClipTowers(synth.synth_clip_state_dict(seed=16))
    assert (t.vision_width, t.vision_layers, t.vision_heads, t.patch) == (768, 12, 12, 32)
    assert (t.text_width, t.text_layers, t.text_heads, t.context_length, t.embed_dim) == (512, 12, 8, 77, 512)
    return t


def test_image_tower_matches_reference_model(golden, towers):
    g = golden("g16_clip_towers")
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(int(g["image_seed"])))
    out = towers.encode_image(img)
    ref = g["image_features"]
    assert out.shape == (2, 512)
    np.testing.assert_allclose(out.numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))


def test_text_tower_matches_reference_model(golden, towers):
    g = golden("g16_clip_towers")
    out = towers.encode_text(torch.from_numpy(g["tokens"]))
    ref = g["text_features"]
    np.testing.assert_allclose(out.numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()))
    # the pooled position is the end-of-text token: changing anything behind it changes nothing
    tok = torch.from_numpy(g["tokens"]).clone()
    eot = int(tok[0].argmax())
    tok[0, eot + 1:] = 7
    np.testing.assert_allclose(towers.encode_text(tok)[0].numpy(), out[0].numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.skipif(not os.path.exists(BPE), reason="CLIP's BPE merges file is only present where the reference tree is")
def test_tokenizer_matches_reference_tokenizer(golden):
    g = golden("g16_clip_towers")
    tk = SimpleTokenizer(BPE)
    got = tk.tokenize([str(p) for p in g["prompts"]])
    assert torch.equal(got, torch.from_numpy(g["tokens"]))
    assert got[0, 0] == 49406 and got[0].max() == 49407 and int(got[4].argmax()) == 76          # truncated: end marker in the last slot
    with pytest.raises(ValueError):
        SimpleTokenizer(None)

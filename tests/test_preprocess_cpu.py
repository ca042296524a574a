"""f4: image-side preprocessing of the image-conditioned driver against fixtures made with the reference's own
mask2bbox / crop_square (tools/make_golden.py g15)."""
import numpy as np
import pytest

from surfd_amd import preprocess as pp

CASES = ["center", "tall_left", "wide_bottom", "corner", "whole", "thin"]


@pytest.mark.parametrize("name", CASES)
def test_mask_crop_matches_reference(golden, name):
    g = golden("g15_image_preprocess")
    img, mask = g[name + "__img"], g[name + "__mask"]
    assert list(pp.mask2bbox(mask)) == [int(v) for v in g[name + "__bbox"]]
    clean, comp = pp.masked_crops(img, mask, r=0.7)
    assert clean.size == (256, 256) and comp.size == (256, 256)
    import hashlib
    for tag, im in (("clean", clean), ("comp", comp)):
        a = np.ascontiguousarray(np.array(im))
        np.testing.assert_array_equal(a[::8, ::8], g[f"{name}__{tag}_sub"])
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(g[f"{name}__{tag}_sha256"])


def test_empty_mask_and_clip_tensor():
    with pytest.raises(IndexError):
        pp.mask2bbox(np.zeros((8, 8), bool))
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    t = pp.clip_image_tensor(img, 224)
    assert tuple(t.shape) == (3, 224, 224)
    # a constant image stays constant under the normalisation + resize: (v / 255 - mean) / std per channel
    c = pp.clip_image_tensor(np.full((256, 256, 3), 128, np.uint8), 224)
    for k in range(3):
        np.testing.assert_allclose(c[k].numpy(), (128 / 255 - pp.CLIP_MEAN[k]) / pp.CLIP_STD[k], rtol=0, atol=1e-6)

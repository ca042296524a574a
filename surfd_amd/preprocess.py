"""Image-side preprocessing of the image-conditioned driver: what sample/generate_image.py does to the
photo and its mask before the CLIP image tower sees them.

  mask2bbox(mask)                 <- data_loaders/dataset.py:19-26   (x0, y0, x1, y1) of the mask's support
  crop_square(img, bbox, ...)     <- data_loaders/dataset.py:29-77   square crop around the box, edge-padded, 256 x 256
  masked_crops(img, mask, r)      <- sample/generate_image.py:92-107 the "clean" (object on black) and "comp" crops
  clip_image_tensor(img, n_px)    <- data_loaders/dataset.py:87-93 (_transform_rgb): ToTensor, Normalize, Resize
  sketch_clip_tensor(img, n_px)   <- sample/generate_sketch.py:30-37 (_transform): Resize(n_px, BICUBIC) of the SHORT side,
                                     CenterCrop(n_px), RGB, ToTensor, Normalize

Host-side numpy / PIL, outside the hot path.  mask2bbox / crop_square / masked_crops are pinned by fixtures made with the
reference's own functions (tests/golden/g15_image_preprocess.npz); the final Resize is torchvision's in the reference
(package absent here): restated as the antialiased bilinear interpolation torchvision applies to tensors — unpinned.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def mask2bbox(mask: np.ndarray) -> Tuple[int, int, int, int]:
    """(x0, y0, x1, y1): first / last column and row that contain a set pixel (inclusive)."""
    mask = np.asarray(mask)
    ys = np.flatnonzero(mask.any(axis=1))
    xs = np.flatnonzero(mask.any(axis=0))
    if len(ys) == 0:
        raise IndexError("mask2bbox: the mask is empty")          # the reference fails on the same input (index -1 of an empty array)
    return int(xs[0]), int(ys[0]), int(xs[-1]), int(ys[-1])


def crop_square(img: np.ndarray, bbox, img_size_h: int = 256, img_size_w: int = 256):
    """Square window of side max(box width, box height) centred on the box, clipped to the image and padded back to a
    square by repeating the border pixels, then resized (PIL's default filter, as the reference) -> PIL.Image."""
    from PIL import Image
    h, w = img.shape[:2]
    x0, y0, x1, y1 = bbox
    side = max(x1 - x0, y1 - y0)
    cx, cy = (x0 + x1) * 0.5, (y0 + y1) * 0.5
    left, right = int(cx - side * 0.5), int(cx + side * 0.5)
    top, bottom = int(cy - side * 0.5), int(cy + side * 0.5)
    pad_l, pad_t = max(0, -left), max(0, -top)
    left, top = max(left, 0), max(top, 0)
    pad_r = pad_b = 0
    if right >= w:
        pad_r, right = right - w + 1, w - 1
    if bottom >= h:
        pad_b, bottom = bottom - h + 1, h - 1
    window = np.pad(img[top:bottom + 1, left:right + 1], ((pad_t, pad_b), (pad_l, pad_r), (0, 0)), mode="edge")
    return Image.fromarray(window).resize((img_size_w, img_size_h))


def masked_crops(img: np.ndarray, mask: np.ndarray, r: float = 0.7):
    """(clean, comp): the object on black — what the image tower is fed — and the object over a whitened background
    (r * 255 + (1 - r) * img outside the mask), both as 256 x 256 square crops around the mask's box."""
    img = np.asarray(img)
    m = np.asarray(mask)[:, :, None]
    bbox = list(mask2bbox(mask))
    comp = img * m + (1 - m) * (r * 255 + (1 - r) * img)
    clean = img * m
    return crop_square(clean.astype(np.uint8), bbox), crop_square(comp.astype(np.uint8), bbox)


def clip_image_tensor(img, n_px: int = 224):
    """uint8 HWC image (PIL or array) -> float32 [3, n_px, n_px]: scale to [0, 1], CLIP's per-channel normalisation, then
    resize (the reference's transform order: Normalize before Resize)."""
    import torch
    import torch.nn.functional as F
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    a = (a - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)
    return F.interpolate(a[None], size=(n_px, n_px), mode="bilinear", antialias=True, align_corners=False)[0]


def resize_short_side(img, n_px: int):
    """torchvision.transforms.Resize(n_px, BICUBIC) on a PIL image: the shorter side becomes n_px, the longer one
    int(n_px * long / short) (torchvision's _compute_resized_output_size), through PIL's own bicubic filter — which is
    what torchvision calls for PIL inputs."""
    from PIL import Image
    w, h = img.size
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        return img
    if w <= h:
        nw, nh = n_px, int(n_px * h / w)
    else:
        nw, nh = int(n_px * w / h), n_px
    return img.resize((nw, nh), Image.BICUBIC)


def center_crop(img, n_px: int):
    """torchvision.transforms.CenterCrop(n_px) on a PIL image: box at int(round((side - n_px) / 2)); an image smaller than
    the crop is first padded with zeros, centred (left / top get the smaller half) like torchvision does."""
    from PIL import Image
    w, h = img.size
    if w < n_px or h < n_px:
        pl, pt = max((n_px - w) // 2, 0), max((n_px - h) // 2, 0)
        canvas = Image.new(img.mode, (max(w, n_px), max(h, n_px)))
        canvas.paste(img, (pl, pt))
        img = canvas
        w, h = img.size
    top, left = int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))
    return img.crop((left, top, left + n_px, top + n_px))


def sketch_clip_tensor(img, n_px: int = 224):
    """PIL image of any size / mode -> float32 [3, n_px, n_px], the sketch driver's transform
    (sample/generate_sketch.py:30-37): resize (short side, bicubic) -> centre crop -> RGB -> [0, 1] -> CLIP normalisation."""
    import torch
    img = center_crop(resize_short_side(img, n_px), n_px).convert("RGB")
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return (a - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)

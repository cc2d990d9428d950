"""Oracle: the evaluation image transform of the reference, restated in numpy.  TEST INFRASTRUCTURE.

Follows /root/reference/mvuld/data/build.py:146-168 (`build_transform`, not-training branch, `TEST.CROP` False):
    transforms.Resize((S, S), interpolation=bicubic) -> transforms.ToTensor() -> transforms.Normalize(IMAGENET_DEFAULT_MEAN, _STD)
applied to the PIL RGB image `data_list.py` opens for every function.  The arithmetic lives in two third-party dependencies that
are not part of /root/reference:
  * torchvision's `Resize` on a PIL image is `img.resize((S, S), PIL.Image.BICUBIC)`;
  * Pillow's resize (`src/libImaging/Resample.c`, unchanged since Pillow 7; pinned here against the installed Pillow, 12.2.0) is a
    two-pass (horizontal, then vertical) separable convolution on 8-bit channels with 22-bit fixed-point coefficients and a
    rounding to uint8 after EACH pass; the bicubic kernel uses a = -0.5 and support 2, stretched by the scale when shrinking
    (antialiasing).
This file restates `precompute_coeffs`, `normalize_coeffs_8bpc` and the two 8bpc passes bit for bit; tests/ pin it against
PIL itself on random images (exact uint8 equality), and the HIP kernels against it.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c:precompute_coeffs (box = the whole axis) + normalize_coeffs_8bpc.
    -> (bounds int32 [out, 2] = (first input index, tap count), kk int32 [out, ksize], ksize)"""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)          # left-to-right double sum, as the C loop
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One 8bpc pass along `axis` of an [H, W, C] uint8 image: ss = 2^21 + sum(pixel * k); out = clip8(ss >> 22)."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)           # [n_in, other, C]
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL.Image.resize((out_w, out_h), BICUBIC) of an [H, W, 3] uint8 RGB image (ImagingResampleInner: the horizontal pass runs
    first and only where the width changes, the vertical one only where the height changes)."""
    h, w, _ = img.shape
    x = img
    if out_w != w:
        bh, kh, _ = precompute_coeffs(w, out_w)
        x = _pass(x, bh, kh, axis=1)
    if out_h != h:
        bv, kv, _ = precompute_coeffs(h, out_h)
        x = _pass(x, bv, kv, axis=0)
    return x


def to_tensor_normalize(img_u8: np.ndarray, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD) -> np.ndarray:
    """ToTensor (HWC uint8 -> CHW float32 / 255) then Normalize ((x - mean) / std), in float32 like torchvision."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (x - m) / s


def eval_transform(img_u8: np.ndarray, size: int) -> np.ndarray:
    """[H, W, 3] uint8 -> [3, size, size] float32: the reference's evaluation transform (build.py:146-168)."""
    return to_tensor_normalize(resize_bicubic_u8(img_u8, size, size))

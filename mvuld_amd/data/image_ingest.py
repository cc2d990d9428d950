"""Image ingestion on the device: the reference's evaluation transform (data/build.py:146-168 -- PIL bicubic Resize((S, S)), ToTensor,
Normalize(IMAGENET mean / std)) applied to decoded uint8 RGB images that already sit in HBM, bit-exact with Pillow's resampler.

The host's share is what Pillow computes once per (input size, output size) pair: the tap tables of the separable filter
(Resample.c: precompute_coeffs + normalize_coeffs_8bpc; bicubic, a = -0.5, support 2 stretched by the scale when shrinking, 22-bit
fixed point).  They are cached per size pair; the pixels never visit the host.  Replaces, per function, PIL resize + ToTensor +
Normalize on the loader thread and the H2D copy of a float image (2.4 MB) by an H2D copy of the raw bytes (H x W x 3).
"""
import math

import numpy as np
import torch

from .. import hip
from ..hip import call, ptr

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_bicubic_taps(in_size, out_size):
    """Pillow's tap table of one axis: (bounds int32 [out, 2] = first input index and tap count, kk int32 [out, ksize], ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    one = 1 << _PRECISION_BITS
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = [_bicubic((x + lo - center + 0.5) * inv) for x in range(n)]
        total = 0.0
        for v in w:                       # the C loop's summation order
            total += v
        for x in range(n):
            k = w[x] / total if total != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
        bounds[xx] = (lo, n)
    return bounds, kk, ksize


class DeviceImageTransform:
    """images uint8 [B, H, W, 3] (or [H, W, 3]) on the GPU -> [B, 3, S, S] float32 (or bf16), as build_transform(is_train=False) would
    have produced from the same pixels on the host."""

    def __init__(self, size=448, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, out_dtype=torch.float32):
        self.size, self.mean, self.std, self.out_dtype = int(size), tuple(mean), tuple(std), out_dtype
        self._taps = {}

    def _tables(self, n_in, device):
        key = (n_in, device)
        t = self._taps.get(key)
        if t is None:
            b, k, ks = pillow_bicubic_taps(n_in, self.size)
            t = self._taps[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
        return t

    def __call__(self, images, return_u8=False):
        hip.require_gpu(images)
        if images.dim() == 3:
            images = images[None]
        assert images.dtype == torch.uint8 and images.shape[-1] == 3 and images.is_contiguous(), "uint8 [B, H, W, 3] RGB expected"
        B, H, W, _ = images.shape
        S = self.size
        bv, kv, ksv = self._tables(H, images.device)
        bh = kh = tmp = None
        ksh = 0
        if W != S:                         # Pillow runs the horizontal pass only where the width changes
            bh, kh, ksh = self._tables(W, images.device)
            tmp = torch.empty((B, H, S, 3), dtype=torch.uint8, device=images.device)
        out = torch.empty((B, 3, S, S), dtype=self.out_dtype, device=images.device)
        u8 = torch.empty((B, S, S, 3), dtype=torch.uint8, device=images.device) if return_u8 else None
        call("image_resize_bicubic_normalize", ptr(images), B, H, W, ptr(bh), ptr(kh), ksh, ptr(bv), ptr(kv), ksv, S, S, ptr(tmp), ptr(out),
             hip.F32 if self.out_dtype == torch.float32 else hip.BF16, ptr(u8), *[float(np.float32(m)) for m in self.mean],
             *[float(np.float32(s)) for s in self.std])
        return (out, u8) if return_u8 else out

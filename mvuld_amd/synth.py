"""Deterministic synthetic weights and inputs (no RNG-version dependence).

Every value is a pure function of (tensor name, flat element index): a 64-bit
splitmix hash mapped to a uniform float in [-1, 1).  The golden-vector generator
(tests/golden/make_golden.py, run in the build container next to the reference)
and the tests / bench on the GPU box regenerate bit-identical tensors from it,
so fixtures hold only outputs.

Why not the reference's own init: SwinV2 zero-initialises every res-post-norm
LayerNorm (swin_transformer_v2.py:447-452) and Rs_GCN zero-initialises its
residual BatchNorm (Rs_GCN.py:33-34), which makes a fresh model an identity
through all attention / FFN / GCN blocks; synthetic
weights here are non-zero everywhere.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def name_seed(name: str, salt: int = 0) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    h2 = zlib.crc32((name + "#").encode()) & 0xFFFFFFFF
    return np.uint64(((h << 32) | h2) ^ (salt * 0x9E3779B1 & 0xFFFFFFFF))


def unit(name: str, n: int, salt: int = 0) -> np.ndarray:
    """n float32 values uniform in [-1, 1), a pure function of (name, salt, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = _splitmix(idx * np.uint64(0xD1342543DE82EF95) + name_seed(name, salt))
    top = (z >> np.uint64(40)).astype(np.float64)          # 24 bits
    return (top / float(1 << 23) - 1.0).astype(np.float32)


def tensor(name, shape, lo=-1.0, hi=1.0, salt=0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit(name, n, salt)
    v = (u * np.float32(0.5) + np.float32(0.5)) * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def ints(name, shape, lo, hi, salt=0) -> torch.Tensor:
    """int64 uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = unit(name, n, salt).astype(np.float64) * 0.5 + 0.5
    v = np.minimum((u * (hi - lo)).astype(np.int64) + lo, hi - 1)
    return torch.from_numpy(v).reshape(tuple(shape))


def synth_param(name: str, shape, salt=0) -> torch.Tensor:
    """Synthetic value for a parameter/buffer chosen by its reference name."""
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.int64)
    if leaf == "running_mean":
        return tensor(name, shape, -0.1, 0.1, salt)
    if leaf == "running_var":
        return tensor(name, shape, 0.8, 1.2, salt)
    if leaf == "logit_scale":
        return tensor(name, shape, 2.0, 2.6, salt)          # around ln(10)
    if leaf in ("attn_l", "attn_r"):
        return tensor(name, shape, -0.1, 0.1, salt)
    if "embeddings" in name and leaf == "weight" and len(shape) == 2 and "LayerNorm" not in name:
        return tensor(name, shape, -0.05, 0.05, salt)
    if len(shape) <= 1:
        # every 1-D "weight" on this path is a LayerNorm / BatchNorm scale
        if leaf == "weight" and ".W.1." in name:
            # Rs_GCN's residual BatchNorm scale: zero at init in the reference (Rs_GCN.py:33), small after training.
            # Non-zero so the GCN kernels are exercised, small so the 8-block residual chain stays well conditioned.
            return tensor(name, shape, 0.05, 0.15, salt)
        if leaf == "weight":
            return tensor(name, shape, 0.75, 1.25, salt)
        return tensor(name, shape, -0.05, 0.05, salt)      # biases, q_bias, v_bias, norm biases
    fan_in = int(np.prod(shape[1:]))
    a = 1.0 / np.sqrt(fan_in)
    return tensor(name, shape, -a, a, salt)


def synth_state_dict(shapes: dict, salt=0) -> dict:
    return {k: synth_param(k, s, salt) for k, s in shapes.items()}

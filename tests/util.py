"""Shared helpers for the parity tests."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from mvuld_amd import synth  # noqa: E402


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def load_synth_into(module, prefix="", skip=("relative_coords_table",)):
    """Fill every parameter / BN buffer of `module` with its synthetic value (by reference key name)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith(skip):
            continue
        new[k] = synth.synth_param(prefix + k, tuple(v.shape)).to(v.dtype)
    missing = module.load_state_dict(new, strict=False)
    return {prefix + k: v for k, v in new.items()}, missing


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 3e-2

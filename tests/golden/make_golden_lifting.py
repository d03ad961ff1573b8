#!/usr/bin/env python3
"""Regenerates tests/golden/lifting_golden.npz.  Run in the BUILD container only (needs /root/reference):
    python tests/golden/make_golden_lifting.py

Pins the legacy lifting transforms (CustomTransform, /root/reference/main/transforms/custom_transforms.py:14-117)
against the REFERENCE's own arithmetic: the sub-package main/transforms/wavelets/ needs only numpy, pandas and torch, so it
is imported here by file path (custom_transforms.py itself is not importable: it pulls pywt / pytorch_wavelets /
torchvision at module level) and `fast_haar_2d_op` / `fast_cdf97_2d_op` are run on seeded inputs.  The multi-level
cascade and the padding rule of HaarLifting / Cdf97Lifting (:18-24, :42-47) are applied around those reference calls.
Fixtures are data only: inputs (as seeds + shapes) and expected outputs.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def load_reference_wavelets():
    pkg_dir = os.path.join(REF, "main", "transforms", "wavelets")
    pkg = types.ModuleType("refwavelets")
    pkg.__path__ = [pkg_dir]
    sys.modules["refwavelets"] = pkg
    for name in ("utils", "haar", "cdf_97"):
        spec = importlib.util.spec_from_file_location(f"refwavelets.{name}", os.path.join(pkg_dir, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refwavelets.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["refwavelets.haar"].fast_haar_2d_op, sys.modules["refwavelets.cdf_97"].fast_cdf97_2d_op


CASES = [  # name, basis, shape [N,C,H,W], levels, seed
    ("haar_8x12", "haar", (1, 3, 8, 12), 1, 1),
    ("haar_64x48_l2", "haar", (2, 3, 64, 48), 2, 2),
    ("haar_odd_7x9", "haar", (1, 2, 7, 9), 1, 3),            # padded to 8 x 10 by HaarLifting.forward_one
    ("haar_224", "haar", (1, 1, 224, 224), 3, 4),
    ("cdf97_8x12", "cdf97", (1, 3, 8, 12), 1, 5),
    ("cdf97_64x48_l2", "cdf97", (2, 3, 64, 48), 2, 6),
    ("cdf97_pad_10x6", "cdf97", (1, 1, 10, 6), 1, 7),        # padded to 12 x 8 (multiple of 4)
    ("cdf97_224", "cdf97", (1, 1, 224, 224), 2, 8),
]


def main():
    haar2d, cdf2d = load_reference_wavelets()
    out = {}
    for name, basis, shape, levels, seed in CASES:
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(shape, generator=g)
        out[f"{name}/seed"] = np.array(seed)
        out[f"{name}/shape"] = np.array(shape)
        cur = x
        for lev in range(levels):
            h, w = cur.shape[-2:]
            if basis == "haar":
                cur = F.pad(cur, (0, w % 2, 0, h % 2))
                ll, lh, hl, hh = haar2d(cur.clone())
            else:
                cur = F.pad(cur, (0, (4 - w % 4) % 4, 0, (4 - h % 4) % 4))
                ll, lh, hl, hh = cdf2d(cur.clone())
            out[f"{name}/l{lev}/ll"] = ll.numpy().copy()
            out[f"{name}/l{lev}/hi"] = torch.stack([lh, hl, hh], dim=-3).numpy().copy()
            cur = ll.clone()
    path = os.path.join(HERE, "lifting_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()

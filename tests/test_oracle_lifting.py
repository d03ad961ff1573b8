"""CPU: the numpy restatement of the legacy lifting transforms against vectors produced by the REFERENCE's own
fast_haar_2d_op / fast_cdf97_2d_op (tests/golden/lifting_golden.npz, made by tests/golden/make_golden_lifting.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import lifting_np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lifting_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def cases(gold):
    return sorted({k.split("/")[0] for k in gold.files})


def test_restatement_reproduces_the_reference_bit_for_bit(gold):
    names = cases(gold)
    assert len(names) == 8
    for name in names:
        basis = "haar" if name.startswith("haar") else "cdf97"
        shape, seed = tuple(gold[f"{name}/shape"]), int(gold[f"{name}/seed"])
        x = torch.randn(shape, generator=torch.Generator().manual_seed(seed)).numpy()
        levels = len([k for k in gold.files if k.startswith(name + "/l") and k.endswith("/ll")])
        approx, details = lifting_np.lifting_levels(x, basis, levels)
        for lev in range(levels):
            assert np.array_equal(approx[lev], gold[f"{name}/l{lev}/ll"]), (name, lev, "ll")
            assert np.array_equal(details[lev], gold[f"{name}/l{lev}/hi"]), (name, lev, "hi")


def test_custom_transform_output_layouts():
    x = np.random.default_rng(0).standard_normal((3, 16, 24)).astype(np.float32)
    y = lifting_np.custom_transform(x, decompose_levels=2, basis="haar")
    assert y.shape == (3, 4, 4, 6)                                   # [C, LL|LH|HL|HH, H/4, W/4]
    assert lifting_np.custom_transform(x, 2, "haar", ll_only=True).shape == (3, 4, 6)
    assert lifting_np.custom_transform(x, 1, "cdf97", coarse_only=False).shape == (3, 4, 8, 12)
    with pytest.raises(NotImplementedError):
        lifting_np.custom_transform(x, 2, "haar", coarse_only=False)
    # haar: LL of a constant image c is c (scales v6 keep the data range), details vanish
    c = np.full((1, 8, 8), 0.75, np.float32)
    z = lifting_np.custom_transform(c, 1, "haar")
    assert np.allclose(z[0, 0], 0.75, atol=1e-6) and np.abs(z[0, 1:]).max() < 1e-6

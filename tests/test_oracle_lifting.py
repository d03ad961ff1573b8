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


def test_haar_swt_level1_is_anchored_to_reference_made_lifting_vectors(gold):
    """Cross-check of the (PyWavelets-unpinned) SWT restatement against numbers the REFERENCE produced: the decimated
    Haar transform is the stationary one sampled at even shifts.  With the reference's lifting scales
    (LL * 1/2, LH, HL * 1, HH * sqrt 2; sign of the detail filters flipped: d = (odd - even)/sqrt 2)
        cA[2i,2j] = 2 LL[i,j],  cH[2i,2j] = -LH[i,j],  cV[2i,2j] = -HL[i,j],  cD[2i,2j] = HH[i,j] / sqrt 2."""
    from oracle import swt_np
    name = "haar_224"
    shape, seed = tuple(gold[f"{name}/shape"]), int(gold[f"{name}/seed"])
    x = torch.randn(shape, generator=torch.Generator().manual_seed(seed)).numpy()[0, 0]
    swt = swt_np.swt2_level_n(x, "haar", 1)                       # [4, 224, 224]: cA, cH, cV, cD
    ll, hi = gold[f"{name}/l0/ll"][0, 0], gold[f"{name}/l0/hi"][0, 0]
    ev = swt[:, 0::2, 0::2]
    tol = 2e-6 * np.abs(x).max()
    assert np.abs(ev[0] - 2 * ll).max() < tol
    assert np.abs(ev[1] + hi[0]).max() < tol
    assert np.abs(ev[2] + hi[1]).max() < tol
    assert np.abs(ev[3] - hi[2] / np.sqrt(2)).max() < tol

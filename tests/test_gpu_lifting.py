"""GPU: the legacy lifting transforms (wv_lifting2d_forward, CustomTransform) against vectors produced by the
REFERENCE's own fast_haar_2d_op / fast_cdf97_2d_op (tests/golden/lifting_golden.npz) -- bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import lifting_np

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "lifting_golden.npz")


def test_lifting_kernel_is_bit_identical_to_the_reference():
    from wvhash.transforms import Cdf97Lifting, HaarLifting
    gold = np.load(GOLD)
    names = sorted({k.split("/")[0] for k in gold.files})
    for name in names:
        basis = "haar" if name.startswith("haar") else "cdf97"
        shape, seed = tuple(gold[f"{name}/shape"]), int(gold[f"{name}/seed"])
        x = torch.randn(shape, generator=torch.Generator().manual_seed(seed))
        levels = len([k for k in gold.files if k.startswith(name + "/l") and k.endswith("/ll")])
        mod = (HaarLifting if basis == "haar" else Cdf97Lifting)(n_levels=levels)
        approx, details = mod(x.cuda())
        for lev in range(levels):
            assert np.array_equal(approx[lev].cpu().numpy(), gold[f"{name}/l{lev}/ll"]), (name, lev, "ll")
            assert np.array_equal(details[lev].cpu().numpy(), gold[f"{name}/l{lev}/hi"]), (name, lev, "hi")


@pytest.mark.parametrize("basis,levels,shape", [("haar", 1, (3, 224, 224)), ("haar", 3, (2, 3, 100, 75)), ("cdf97", 1, (3, 448, 448)),
                                                ("cdf97", 2, (1, 3, 50, 30))])
def test_custom_transform_matches_oracle_and_keeps_the_callers_device(basis, levels, shape):
    from wvhash.transforms import CustomTransform
    x = torch.randn(shape, generator=torch.Generator().manual_seed(5))
    y = CustomTransform(decompose_levels=levels, basis=basis)(x)            # CPU in -> CPU out (worker-style call)
    assert not y.is_cuda
    ref = lifting_np.custom_transform(x.numpy(), levels, basis)
    assert y.shape == ref.shape and np.array_equal(y.numpy(), ref)
    ll = CustomTransform(decompose_levels=levels, basis=basis, ll_only=True)(x.cuda())
    assert ll.is_cuda and np.array_equal(ll.cpu().numpy(), lifting_np.custom_transform(x.numpy(), levels, basis, ll_only=True))


def test_full_subbands_level1_and_errors_and_resize():
    from wvhash import _lib
    from wvhash.transforms import CustomTransform, ResizeSubBands
    x = torch.randn(3, 32, 48, generator=torch.Generator().manual_seed(6))
    y = CustomTransform(decompose_levels=1, basis="cdf97", coarse_only=False)(x)
    assert np.array_equal(y.numpy(), lifting_np.custom_transform(x.numpy(), 1, "cdf97", coarse_only=False))
    with pytest.raises(NotImplementedError):
        CustomTransform(decompose_levels=2, basis="haar", coarse_only=False)(x)
    r = ResizeSubBands(20)(y)                                                # [3, 4, 16, 24] -> shorter side 20
    assert tuple(r.shape) == (3, 4, 20, 30)
    ref = torch.nn.functional.interpolate(y.reshape(12, 1, 16, 24), size=(20, 30), mode="bilinear", align_corners=False, antialias=True)
    assert torch.equal(r, ref.reshape(3, 4, 20, 30))
    lib = _lib.require_gpu()
    t = torch.zeros(64, device="cuda")
    assert lib.wv_lifting2d_forward(_lib.ptr(t), 1, 7, 8, 0, _lib.ptr(t), _lib.ptr(t), None, 0, None) == -22   # odd H
    assert lib.wv_lifting2d_forward(_lib.ptr(t), 1, 8, 8, 0, _lib.ptr(t), _lib.ptr(t), None, 0, None) == -12   # no workspace

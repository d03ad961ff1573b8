"""GPU: wv_dwt2d_forward (DWTTransform) against the numpy restatement of pywt.wavedec2's coarsest level."""
import numpy as np
import pytest
import torch

from oracle import swt_np

pytestmark = pytest.mark.gpu

TOL = 2e-6   # fp32 taps, fma vs mul+add accumulation, at most 3 levels * 2 axes * 10 taps


@pytest.mark.parametrize("wavelet,level,shape", [
    ("haar", 1, (2, 3, 32, 32)), ("haar", 2, (1, 3, 224, 224)), ("db2", 1, (2, 3, 40, 56)),
    ("db2", 3, (1, 3, 224, 224)), ("db4", 2, (1, 2, 37, 51)), ("bior4.4", 1, (1, 1, 64, 48)),
    ("db4", 1, (1, 1, 3, 5)),
])
def test_dwt_matches_oracle(wavelet, level, shape):
    from wvhash.transforms import dwt2d
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = dwt2d(torch.from_numpy(x).cuda(), wavelet, level).cpu().numpy()
    for b in range(shape[0]):
        for c in range(shape[1]):
            exp = swt_np.wavedec2_coarsest(x[b, c].astype(np.float32) / np.float32(255), wavelet, level)
            assert got[b, c].shape == exp.shape
            np.testing.assert_allclose(got[b, c], exp, atol=TOL * 2 ** level, rtol=0)


def test_dwt_layouts_and_float_input():
    from wvhash.transforms import dwt2d
    rng = np.random.default_rng(4)
    x = rng.random((2, 3, 48, 64), dtype=np.float32)
    a = dwt2d(torch.from_numpy(x).cuda(), "db2", 2)
    b = dwt2d(torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda(), "db2", 2, channels_last=True)
    assert torch.equal(a, b)
    exp = swt_np.wavedec2_coarsest(x[1, 2], "db2", 2)
    np.testing.assert_allclose(a[1, 2].cpu().numpy(), exp, atol=8e-6, rtol=0)


def test_dwt_transform_class_and_errors():
    from wvhash import _lib
    from wvhash.transforms import DWTTransform
    img = np.random.default_rng(5).integers(0, 256, size=(30, 45, 3), dtype=np.uint8)   # fix_size -> 32 x 48
    t = DWTTransform(level=2, wavelet="haar", device="cuda")
    from PIL import Image
    out = t(Image.fromarray(img))
    assert out.shape == (3, 4, 8, 12) and out.dtype == torch.float32
    lib = _lib.require_gpu()
    x = torch.zeros((1, 1, 8, 8), dtype=torch.uint8, device="cuda")
    o = torch.empty((1, 1, 4, 4, 4), device="cuda")
    f = _lib.host_floats([0.5, 0.5])
    assert lib.wv_dwt2d_forward(_lib.ptr(x), 0, 0, _lib.ptr(o), 1, 1, 8, 8, 1, f, f, 2, None, 0, None) == -12
    assert lib.wv_dwt2d_forward(_lib.ptr(x), 0, 0, _lib.ptr(o), 1, 1, 8, 8, 0, f, f, 2, None, 0, None) == -22

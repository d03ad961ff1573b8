"""CPU: the numpy restatement of DWTTransform's arithmetic (pywt.wavedec2, mode 'symmetric';
reference custom_transforms.py:197-201) against analytic known answers.  Parity unpinned against
PyWavelets itself (not installed here, no fixtures in the reference)."""
import numpy as np
import pytest

from oracle import swt_np


def test_haar_is_2x2_block_transform():
    rng = np.random.default_rng(1)
    x = rng.random((16, 24), dtype=np.float32)
    y = swt_np.wavedec2_coarsest(x, "haar", 1)
    b = x.reshape(8, 2, 12, 2).astype(np.float64)
    x00, x01, x10, x11 = b[:, 0, :, 0], b[:, 0, :, 1], b[:, 1, :, 0], b[:, 1, :, 1]
    # pywt dec_hi for haar = [-s, s]:  y[o] = hi[0]*x[2o+1] + hi[1]*x[2o]  = s*(x[2o] - x[2o+1])
    np.testing.assert_allclose(y[0], (x00 + x01 + x10 + x11) / 2, atol=1e-6)
    np.testing.assert_allclose(y[1], ((x00 + x01) - (x10 + x11)) / 2, atol=1e-6)   # cH: detail along axis 0
    np.testing.assert_allclose(y[2], ((x00 + x10) - (x01 + x11)) / 2, atol=1e-6)   # cV: detail along axis 1
    np.testing.assert_allclose(y[3], (x00 - x01 - x10 + x11) / 2, atol=1e-6)


@pytest.mark.parametrize("wavelet,level", [("haar", 1), ("haar", 3), ("db2", 1), ("db2", 2), ("db4", 2), ("bior4.4", 1)])
def test_shape_and_constant_image(wavelet, level):
    lo, _ = swt_np.filters(wavelet)
    H, W = 40, 56
    h, w = H, W
    for _ in range(level):
        h, w = (h + len(lo) - 1) // 2, (w + len(lo) - 1) // 2
    c = np.full((H, W), 0.25, np.float32)
    y = swt_np.wavedec2_coarsest(c, wavelet, level)
    assert y.shape == (4, h, w)
    # symmetric extension of a constant is constant: cA = 2^level * c everywhere, details vanish
    np.testing.assert_allclose(y[0], 0.25 * 2 ** level, rtol=2e-6)
    assert np.abs(y[1:]).max() < 1e-6


def test_linearity_and_interior_matches_plain_convolution():
    rng = np.random.default_rng(2)
    a, b = rng.random((32, 32), dtype=np.float32), rng.random((32, 32), dtype=np.float32)
    ya, yb = swt_np.wavedec2_coarsest(a, "db2", 2), swt_np.wavedec2_coarsest(b, "db2", 2)
    np.testing.assert_allclose(swt_np.wavedec2_coarsest(a + 2 * b, "db2", 2), ya + 2 * yb, atol=2e-5)
    # away from the borders the extension is irrelevant: y[o] = sum_j f[j] x[2o+1-j]
    lo, _ = swt_np.filters("db2")
    row = a[5].astype(np.float64)
    full = np.convolve(row, lo)[1::2]            # full[i] = sum_j lo[j] row[i-j], sampled at i = 1, 3, ...
    got = swt_np.dwt_axis(a, lo, 1, np.float64)[5]
    np.testing.assert_allclose(got[2:-2], full[2:len(got) - 2], atol=1e-12)


def test_short_signal_wraps_extension():
    x = np.arange(3, dtype=np.float32).reshape(1, 3)        # shorter than db4's 8 taps
    y = swt_np.dwt_axis(x, swt_np.filters("db4")[0], 1)
    assert y.shape == (1, 5) and np.isfinite(y).all()

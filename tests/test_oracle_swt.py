"""Pins oracle/swt_oracle.c + oracle/swt_np.py (CPU only).

PyWavelets cannot be imported here ("parity unpinned" against pywt itself), so the restatement is
pinned by every known answer the reference and the algorithm offer:
Haar closed form and the value ranges recorded in
/root/reference/studies/results/swt_transform_check_2026-08-12.txt, constant gain 2^n, zero-sum
details, the 4x energy identity, shift equivariance, impulse responses, C == numpy, and the
committed golden vectors.
"""
import hashlib

import numpy as np
import pytest

from oracle import swt_np
from wvhash import synth

CASES = [("haar", 1), ("haar", 3), ("db2", 1), ("db2", 3), ("db4", 1), ("db4", 2), ("bior4.4", 1), ("bior4.4", 2)]


def rand_plane(h, w, seed=0):
    return np.random.default_rng(seed).random((h, w), dtype=np.float32)


@pytest.mark.parametrize("wl,lev", CASES)
def test_c_equals_numpy_bit_exact(wl, lev):
    x = rand_plane(32, 48, 1)
    assert np.array_equal(swt_np.swt2_level_n(x, wl, lev), swt_np.c_swt2_level_n(x, wl, lev))


def test_haar_level1_closed_form():
    x = rand_plane(16, 24, 2)
    y = swt_np.c_swt2_level_n(x, "haar", 1).astype(np.float64)
    x64 = x.astype(np.float64)
    r = lambda a, dy, dx: np.roll(np.roll(a, -dy, 0), -dx, 1)
    x00, x10, x01, x11 = x64, r(x64, 1, 0), r(x64, 0, 1), r(x64, 1, 1)
    np.testing.assert_allclose(y[0], (x00 + x10 + x01 + x11) / 2, atol=2e-7)   # cA
    np.testing.assert_allclose(y[1], (x00 + x01 - x10 - x11) / 2, atol=2e-7)   # cH = 'da'
    np.testing.assert_allclose(y[2], (x00 - x01 + x10 - x11) / 2, atol=2e-7)   # cV = 'ad'
    np.testing.assert_allclose(y[3], (x00 - x01 - x10 + x11) / 2, atol=2e-7)   # cD


def test_recorded_value_ranges_of_reference_run():
    """swt_transform_check_2026-08-12.txt: Haar L1 on a [0,1] image gives LL in [0, 2] (an image
    with pure black and pure white 2x2 blocks reaches both ends exactly), details in [-1, 1]
    with mean ~ 0."""
    img = np.zeros((1, 8, 8, 3), np.uint8)
    img[0, :4] = 255
    y = swt_np.c_transform_batch(img, "haar", 1)[0]
    assert y.shape == (3, 4, 8, 8) and y.dtype == np.float32
    assert y[:, 0].min() == 0.0 and abs(float(y[:, 0].max()) - 2.0) < 3e-7  # fp32 taps: 1.9999999
    assert np.abs(y[:, 1:]).max() <= 1.0
    nat = synth.natural_images(1, 64, 64, seed=3)
    z = swt_np.c_transform_batch(nat, "haar", 1)[0]
    assert 0.0 <= z[:, 0].min() and z[:, 0].max() <= 2.0
    assert abs(z[:, 1:].mean()) < 1e-6


@pytest.mark.parametrize("wl,lev", CASES)
def test_constant_image_gain_and_zero_details(wl, lev):
    x = np.full((16, 32), 0.5, np.float32)
    y = swt_np.c_swt2_level_n(x, wl, lev)
    np.testing.assert_allclose(y[0], 0.5 * 2 ** lev, rtol=2e-6)
    assert np.abs(y[1:]).max() < 2e-6


@pytest.mark.parametrize("wl,lev", CASES)
def test_detail_bands_sum_to_zero(wl, lev):
    y = swt_np.swt2_level_n(rand_plane(32, 32, 4), wl, lev, dtype=np.float64)
    for b in (1, 2, 3):
        assert abs(y[b].sum()) < 1e-8  # bior4.4's tabulated dec_hi sums to -1.4e-12


@pytest.mark.parametrize("wl", ["haar", "db2", "db4"])
@pytest.mark.parametrize("lev", [1, 2, 3])
def test_energy_identity_orthonormal(wl, lev):
    """per 2-D level: |aa|^2 + |da|^2 + |ad|^2 + |dd|^2 = 4 |A_{l-1}|^2."""
    x = rand_plane(32, 32, 5).astype(np.float64)
    prev = x if lev == 1 else swt_np.swt2_level_n(x, wl, lev - 1, dtype=np.float64)[0]
    y = swt_np.swt2_level_n(x, wl, lev, dtype=np.float64)
    assert abs((y ** 2).sum() / (4 * (prev ** 2).sum()) - 1.0) < 1e-12


@pytest.mark.parametrize("wl,lev", CASES)
def test_shift_equivariance(wl, lev):
    x = rand_plane(32, 32, 6)
    y = swt_np.c_swt2_level_n(x, wl, lev)
    ys = swt_np.c_swt2_level_n(np.roll(np.roll(x, 5, 0), -3, 1), wl, lev)
    assert np.array_equal(np.roll(np.roll(y, 5, 1), -3, 2), ys)


@pytest.mark.parametrize("wl,lev", [("db2", 1), ("db2", 2), ("db4", 1), ("bior4.4", 1)])
def test_impulse_response_places_dilated_taps(wl, lev):
    """delta at p -> y[o] = f[m] exactly where o + s(L/2 - m) = p (1-D rule of 8a-1)."""
    lo, hi = swt_np.filters(wl)
    L, s, n, p = len(lo), 1 << (lev - 1), 64, 20
    x = np.zeros(n, np.float64)
    x[p] = 1.0
    if lev == 2:  # level-2 filters act on A_1; test the level-2 pass in isolation
        y = swt_np.atrous_axis(x, hi, s, 0, np.float64)
    else:
        y = swt_np.atrous_axis(x, hi, 1, 0, np.float64)
    expect = np.zeros(n)
    for m in range(L):
        expect[(p - s * (L // 2 - m)) % n] += hi[m]
    np.testing.assert_allclose(y, expect, atol=0)


def test_filter_table_properties():
    for name in ("haar", "db2", "db4"):
        lo, hi = swt_np.filters(name)
        L = len(lo)
        assert abs(lo.sum() - np.sqrt(2)) < 1e-12 and abs((lo ** 2).sum() - 1) < 1e-11
        assert abs(hi.sum()) < 1e-12
        np.testing.assert_allclose(hi, [(-1) ** (k + 1) * lo[L - 1 - k] for k in range(L)], atol=0)
        for shift in range(2, L, 2):
            assert abs((lo[shift:] * lo[:-shift]).sum()) < 1e-11
    lo, hi = swt_np.filters("db2")
    s3 = np.sqrt(3.0)
    np.testing.assert_allclose(lo, np.array([1 - s3, 3 - s3, 3 + s3, 1 + s3]) / (4 * np.sqrt(2)), atol=1e-15)
    lo, hi = swt_np.filters("bior4.4")
    assert abs(lo.sum() - np.sqrt(2)) < 1e-11 and abs(hi.sum()) < 1e-11


def test_rejects_sizes_not_multiple_of_2_pow_level():
    with pytest.raises(ValueError):
        swt_np.swt2_level_n(rand_plane(12, 16), "haar", 3)
    with pytest.raises(ValueError):
        swt_np.c_swt2_level_n(rand_plane(12, 16), "haar", 3)


def test_fix_size_rounds_up_to_multiple():
    assert swt_np.fix_size_shape(224, 224, 3) == (224, 224)
    assert swt_np.fix_size_shape(225, 30, 3) == (232, 32)


def test_rawstack_and_layout():
    img = synth.noise_images(1, 8, 12, seed=9)
    y = swt_np.c_transform_batch(img, mode="raw")[0]
    for c in range(3):
        for b in range(4):
            np.testing.assert_array_equal(y[c, b], img[0, :, :, c].astype(np.float32) / 255.0)
    np.testing.assert_array_equal(y, swt_np.transform_image(img[0], mode="raw"))


def test_golden_vectors(golden_dir):
    g = np.load(f"{golden_dir}/swt_golden.npz")
    names = sorted({k.split("/")[0] for k in g.files if k.endswith("/img")})
    assert len(names) >= 6
    for n in names:
        wl = bytes(g[n + "/wavelet"]).decode()
        lev = int(g[n + "/meta"][0])
        y = swt_np.c_transform_batch(g[n + "/img"], wl, lev)
        assert np.array_equal(y, g[n + "/out"]), n
        np.testing.assert_array_equal(y[0], swt_np.transform_image(g[n + "/img"][0], wl, lev))
    y = swt_np.c_transform_batch(synth.natural_images(1, 224, 224, seed=1234), "db2", 3)
    assert hashlib.sha256(y.tobytes()).digest() == bytes(g["db2_l3_224/sha"])
    np.testing.assert_array_equal(y.reshape(-1)[::9973], g["db2_l3_224/samples"])

"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/wvhash.h declares, validates arguments before touching the GPU, and the Python plugin
surface refuses to run without the HIP path (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import wvhash
from wvhash import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "wvhash.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wv_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/wvhash.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.wv_abi_version() == _lib.ABI_VERSION == 5
    import __graft_entry__                               # build() checks the same constant
    assert "_lib.ABI_VERSION" in open(__graft_entry__.__file__).read()


def test_release_library_never_reads_the_environment():
    """include/wvhash.h promises "no global mutable state": the kernel-selection switches (WV_SWT_PATH, WV_HEAD_FRONT,
    WV_TOPK_V2, ...) exist only in libwvhash_diag.so (csrc/tune_diag.cpp); libwvhash.so does not even import getenv.  Both
    export the same symbols (they are the same objects but one)."""
    import subprocess
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", _lib.DIAG_LIB_PATH], capture_output=True, text=True,
                                      check=True).stdout
    with _lib.diagnostic() as d:
        assert d is not _lib._LIB and all(hasattr(d, s) for s in header_symbols())
    assert _lib.load() is _lib._LIB


def test_argument_validation_happens_on_the_host():
    lib = _lib.load()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    lo = _lib.host_floats([0.7, 0.7])
    rc = lib.wv_swt2d_forward(one, 0, 0, one, 1, 1, 3, 12, 16, 3, lo, lo, 2, None, 0, None)
    assert rc == -22 and b"multiples of 2^level" in lib.wv_last_error()
    rc = lib.wv_hamming_topk(one, one, one, one, 4, 100, 64, 101, 0, one, 1 << 20, None)
    assert rc == -22 and b"k=101" in lib.wv_last_error()
    rc = lib.wv_hamming_topk(one, one, one, one, 4, 100, 256, 10, 0, one, 1 << 20, None)
    assert rc == -22
    rc = lib.wv_hamming_topk(one, one, one, one, 4, 100, 64, 10, 0, one, 8, None)
    assert rc == -12 and b"workspace" in lib.wv_last_error()
    assert lib.wv_hamming_topk_workspace_bytes(4, 25000, 1, 5000) == 98 * 256 * 8
    assert lib.wv_hamming_dist(None, one, one, 10, 1, 10, 1, None) == -22
    assert lib.wv_swt2d_workspace_bytes(2, 3, 224, 224, 3, 4) == 0          # tiled kernel covers it
    assert lib.wv_swt2d_workspace_bytes(2, 3, 30, 30, 1, 2) == 3 * 2 * 3 * 30 * 30 * 4  # W % 4 != 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_product_path_fails_loudly_without_gpu():
    from wvhash.transforms import SWTTransform, swt2d
    from wvhash.engine import CustomCalculator, get_knn
    with pytest.raises(_lib.WvhashUnavailable):
        swt2d(torch.zeros(1, 3, 8, 8, dtype=torch.uint8))
    with pytest.raises(_lib.WvhashUnavailable):
        SWTTransform(1, "haar")(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(_lib.WvhashUnavailable):
        get_knn(torch.ones(4, 16), torch.ones(2, 16), 2, False, distance_metric="hamming")
    calc = CustomCalculator(k=5, distance_metric="hamming", with_faiss=False)
    with pytest.raises(_lib.WvhashUnavailable):
        calc.calculate_maphashing(torch.ones(2, 16), torch.ones(2, 3), torch.ones(9, 16), torch.ones(9, 3), 5)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image-retrieval-wavelet_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle/"
                assert "liboracle" not in src, f


def test_plugin_surface_names_and_repr():
    from wvhash import transforms as T
    for name in ("SWTTransform", "DWTTransform", "RawStackTransform"):
        assert hasattr(T, name)
    assert repr(T.SWTTransform(level=1, wavelet="haar")) == "SWTTransform(shape='C,S,H,W', wavelet=haar, level=1)"
    assert repr(T.RawStackTransform(copies=4)) == "RawStackTransform(shape='C,4,H,W', copies=4)"
    assert repr(T.DWTTransform(level=2)) == "DWTTransform(shape='C,S,H/4,W/4', wavelet=haar, level=2)"


def test_deferred_transform_only_sizes_the_image():
    from PIL import Image
    from wvhash.transforms import SWTTransform
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (30, 45, 3), dtype=np.uint8))
    t = SWTTransform(level=3, wavelet="db2", defer=True)
    out = t(img)
    assert out.dtype == torch.uint8 and tuple(out.shape) == (3, 32, 48)   # next multiples of 8
    same = SWTTransform(level=1, defer=True)(Image.fromarray(np.zeros((224, 224, 3), np.uint8)))
    assert tuple(same.shape) == (3, 224, 224)


def test_wavelet_table_matches_oracle_table():
    from oracle import swt_np
    from wvhash.transforms import get_filters, wavelist
    from wvhash.transforms.wavelets import _COMPUTED_NAMES, _daubechies
    for name in wavelist():
        if name in _COMPUTED_NAMES:
            continue                                  # computed orders: next test
        lo, hi = get_filters(name)
        olo, ohi = swt_np.filters(name)
        np.testing.assert_array_equal(lo, olo)
        np.testing.assert_array_equal(hi, ohi)
    with pytest.raises(ValueError):
        get_filters("nope")
    assert get_filters(([1, 2], [3, 4])) == ([1.0, 2.0], [3.0, 4.0])
    # the routine behind the computed orders reproduces the tabulated (PyWavelets) db2 / db4
    for n, name in ((2, "db2"), (4, "db4")):
        np.testing.assert_allclose(_daubechies(n), get_filters(name)[0], rtol=0, atol=1e-12)


def test_computed_daubechies_orders_are_orthonormal_with_n_vanishing_moments():
    """db3, db5 ... db10 are not in the reference's configs; they are computed (spectral factorisation, extremal phase).
    What defines them: 2N taps, sum sqrt(2), unit energy, orthogonal to their even shifts, N vanishing moments of the
    high-pass filter; db3 also against its published taps.  And they run through the transform's host twin."""
    from PIL import Image
    from wvhash.transforms import SWTTransform, get_filters
    db3 = [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313, 0.8068915093133388,
           0.3326705529509569]
    np.testing.assert_allclose(get_filters("db3")[0], db3, rtol=0, atol=1e-11)
    for n in (3, 5, 6, 7, 8, 9, 10):
        lo, hi = (np.array(v) for v in get_filters(f"db{n}"))
        L = len(lo)
        assert L == 2 * n and abs(lo.sum() - 2 ** 0.5) < 1e-12 and abs((lo * lo).sum() - 1) < 1e-12 and abs(hi.sum()) < 1e-9
        for m in range(1, n):
            assert abs((lo[: L - 2 * m] * lo[2 * m:]).sum()) < 1e-12
        k = np.arange(L, dtype=np.float64)
        for p in range(n):                                          # moments of the high-pass filter, relative to their scale
            assert abs((hi * k ** p).sum()) < 1e-9 * max(1.0, (np.abs(hi) * k ** p).sum()), (n, p)
    img = Image.fromarray(np.random.default_rng(0).integers(0, 256, (32, 32, 3)).astype(np.uint8))
    x = np.asarray(img, np.float64).transpose(2, 0, 1) / 255.0
    for name in ("db3", "db6"):
        out = SWTTransform(level=1, wavelet=name, device="cpu")(img).double().numpy()     # [3, 4, 32, 32]
        # undecimated, orthonormal, periodic: the four level-1 bands carry 4 x the energy of the image
        np.testing.assert_allclose((out ** 2).sum(axis=(1, 2, 3)), 4 * (x ** 2).sum(axis=(1, 2)), rtol=1e-5)


def test_plain_c_program_runs_the_host_twins_without_a_gpu():
    """tests/native/host_smoke.c: gcc -std=c99 on include/wvhash.h (the header is C), linked against libwvhash.so, calling
    only `_cpu` entry points -- closed-form SWT, packing, distances, stable ranking, AP, hit counts, argument validation --
    on this GPU-less box: the host side of the boundary needs neither torch nor HIP headers nor a device."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native"), "host_smoke"], stdout=subprocess.DEVNULL)
    res = subprocess.run([os.path.join(ROOT, "tests", "native", "host_smoke")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "host_smoke ok" in res.stdout, res.stderr

"""oracle/swt_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent numpy restatement of the reference's wavelet transform path:

* ``swt2_level_n``      <- ``pywt.swt2(channel, wavelet, level)[0]`` as called at
  /root/reference/main/transforms/custom_transforms.py:163-166
* ``transform_image``   <- ``BaseWaveletTransform.__call__`` custom_transforms.py:145-157
* ``raw_stack``         <- ``RawStackTransform._apply_wavelet`` custom_transforms.py:184-185
* ``fix_size``          <- ``BaseWaveletTransform.fix_size`` custom_transforms.py:132-139

PyWavelets (third-party C, not vendored / pinned / installed) holds the arithmetic; the rule
restated here is its published periodized a-trous convolution:

    y[o] = sum_{m=0}^{L-1} f[m] * x[(o + 2^(l-1) * (L/2 - m)) mod N]

axis 0 first, then axis 1; bands cA='aa', cH='da', cV='ad', cD='dd' (first letter = axis 0);
coarsest level only.  Accumulation is in float32, tap order m = 0..L-1, no FMA, like the
single-precision C path PyWavelets takes for float32 input.

PARITY STATUS: parity unpinned against PyWavelets itself (see oracle/swt_oracle.c header);
pinned by the analytic known answers in tests/test_oracle_swt.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os

import numpy as np

_S2 = 0.7071067811865476

# Decomposition filters in PyWavelets convention (dec_lo, dec_hi; index 0 first).
# Names are the ones the reference's configs / studies use.
WAVELETS = {
    "haar": ([_S2, _S2], [-_S2, _S2]),
    "db2": (
        [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
        [-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145],
    ),
    "db4": (
        [-0.010597401784997278, 0.032883011666982945, 0.030841381835986965, -0.18703481171888114,
         -0.02798376941698385, 0.6308807679295904, 0.7148465705525415, 0.23037781330885523],
        [-0.23037781330885523, 0.7148465705525415, -0.6308807679295904, -0.02798376941698385,
         0.18703481171888114, 0.030841381835986965, -0.032883011666982945, -0.010597401784997278],
    ),
    "bior4.4": (
        [0.0, 0.03782845550726404, -0.023849465019556843, -0.11062440441843718,
         0.37740285561283066, 0.8526986790088938, 0.37740285561283066, -0.11062440441843718,
         -0.023849465019556843, 0.03782845550726404],
        [0.0, -0.06453888262869706, 0.04068941760916406, 0.41809227322161724,
         -0.7884856164055829, 0.41809227322161724, 0.04068941760916406, -0.06453888262869706,
         0.0, 0.0],
    ),
}
WAVELETS["db1"] = WAVELETS["haar"]
WAVELETS["sym2"] = WAVELETS["db2"]


def filters(wavelet):
    """(dec_lo, dec_hi) as float64 arrays."""
    if isinstance(wavelet, str):
        lo, hi = WAVELETS[wavelet]
    else:
        lo, hi = wavelet
    return np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)


def atrous_axis(x, f, s, axis, dtype=np.float32):
    """One periodized a-trous pass along `axis`; accumulate in `dtype`, tap order 0..L-1."""
    x = np.asarray(x, dtype=dtype)
    f = np.asarray(f, dtype=dtype)
    L = f.shape[0]
    acc = np.zeros_like(x)
    for m in range(L):
        shift = s * (L // 2 - m)
        # value at output o is x[(o + shift) mod N]  ==  roll by -shift
        acc = acc + f[m] * np.roll(x, -shift, axis=axis)
    return acc


def swt2_level_n(plane, wavelet, level, dtype=np.float32):
    """Coarsest-level (cA, cH, cV, cD) stacked [4,H,W] for one [H,W] plane."""
    lo, hi = filters(wavelet)
    a = np.asarray(plane, dtype=dtype)
    H, W = a.shape
    if H % (1 << level) or W % (1 << level):
        raise ValueError("Length of data must be even along the transform axis / divisible by 2**level")
    for l in range(1, level + 1):
        s = 1 << (l - 1)
        ta = atrous_axis(a, lo, s, 0, dtype)
        td = atrous_axis(a, hi, s, 0, dtype)
        aa = atrous_axis(ta, lo, s, 1, dtype)
        ad = atrous_axis(ta, hi, s, 1, dtype)
        da = atrous_axis(td, lo, s, 1, dtype)
        dd = atrous_axis(td, hi, s, 1, dtype)
        a = aa
    return np.stack([aa, da, ad, dd])


def dwt_axis(x, f, axis, dtype=np.float32):
    """One decimating pass, PyWavelets mode 'symmetric':  y[o] = sum_j f[j] * x_ext[2o + 1 - j],
    o = 0 .. floor((N + F - 1) / 2) - 1, half-sample symmetric extension (x[-1] = x[0], x[N] = x[N-1], ...).
    (pywt/_extensions/c/convolution.template.c: downsampling_convolution, step 2.)  Parity unpinned against
    PyWavelets itself, like the SWT restatement."""
    x = np.moveaxis(np.asarray(x, dtype=dtype), axis, 0)
    f = np.asarray(f, dtype=dtype)
    n, F = x.shape[0], f.shape[0]
    no = (n + F - 1) // 2
    period = 2 * n
    out = np.zeros((no,) + x.shape[1:], dtype=dtype)
    for o in range(no):
        acc = np.zeros(x.shape[1:], dtype=dtype)
        for j in range(F):
            i = (2 * o + 1 - j) % period
            i = i if i < n else period - 1 - i
            acc = acc + f[j] * x[i]
        out[o] = acc
    return np.moveaxis(out, 0, axis)


def wavedec2_coarsest(plane, wavelet, level, dtype=np.float32):
    """(cA_n, cH_n, cV_n, cD_n) stacked [4, H', W'] -- what DWTTransform._apply_wavelet returns
    (custom_transforms.py:197-201)."""
    lo, hi = filters(wavelet)
    a = np.asarray(plane, dtype=dtype)
    for _ in range(level):
        ta, td = dwt_axis(a, lo, 0, dtype), dwt_axis(a, hi, 0, dtype)
        aa, ad = dwt_axis(ta, lo, 1, dtype), dwt_axis(ta, hi, 1, dtype)
        da, dd = dwt_axis(td, lo, 1, dtype), dwt_axis(td, hi, 1, dtype)
        a = aa
    return np.stack([aa, da, ad, dd])


def fix_size_shape(w, h, level):
    factor = 2 ** level
    return int(np.ceil(w / factor) * factor), int(np.ceil(h / factor) * factor)


def fix_size(img, level):
    """PIL image -> PIL image resized (BICUBIC) to the next multiple of 2**level per side."""
    from PIL import Image

    w, h = img.size
    nw, nh = fix_size_shape(w, h, level)
    if nw != w or nh != h:
        img = img.resize((nw, nh), resample=Image.BICUBIC)
    return img


def transform_image(img_hwc_u8, wavelet="haar", level=1, mode="swt", copies=4, dtype=np.float32):
    """HWC uint8 (already sized) -> [3, 4, H, W] float32, channel-major then band."""
    img_np = np.asarray(img_hwc_u8).astype(np.float32) / 255.0
    chans = []
    for c in range(3):
        ch = img_np[:, :, c]
        if mode == "raw":
            chans.append(np.stack([ch] * copies))
        else:
            chans.append(swt2_level_n(ch, wavelet, level, dtype))
    return np.stack(chans).astype(dtype)


# ---------------------------------------------------------------------------------------
# ctypes view of the C restatement (oracle/swt_oracle.c), built by oracle/Makefile
# ---------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def c_lib():
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        lib.wvo_swt2_plane_f32.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int, ctypes.c_int, fp]
        lib.wvo_swt2_plane_f32.restype = ctypes.c_int
        lib.wvo_transform_batch_u8.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        lib.wvo_transform_batch_u8.restype = ctypes.c_int
        _CLIB = lib
    return _CLIB


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def c_swt2_level_n(plane, wavelet, level):
    lo, hi = filters(wavelet)
    lo32 = np.ascontiguousarray(lo, dtype=np.float32)
    hi32 = np.ascontiguousarray(hi, dtype=np.float32)
    p = np.ascontiguousarray(plane, dtype=np.float32)
    H, W = p.shape
    out = np.empty((4, H, W), dtype=np.float32)
    rc = c_lib().wvo_swt2_plane_f32(_fptr(p), H, W, _fptr(lo32), _fptr(hi32), len(lo32), level, _fptr(out))
    if rc:
        raise ValueError(f"wvo_swt2_plane_f32 rc={rc}")
    return out


def c_transform_batch(imgs_bhwc_u8, wavelet="haar", level=1, mode="swt"):
    lo, hi = filters(wavelet)
    lo32 = np.ascontiguousarray(lo, dtype=np.float32)
    hi32 = np.ascontiguousarray(hi, dtype=np.float32)
    x = np.ascontiguousarray(imgs_bhwc_u8, dtype=np.uint8)
    B, H, W, C = x.shape
    assert C == 3
    out = np.empty((B, 3, 4, H, W), dtype=np.float32)
    rc = c_lib().wvo_transform_batch_u8(x.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B, H, W,
                                        _fptr(lo32), _fptr(hi32), len(lo32), level,
                                        1 if mode == "raw" else 0, _fptr(out))
    if rc:
        raise ValueError(f"wvo_transform_batch_u8 rc={rc}")
    return out

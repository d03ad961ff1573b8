"""Batched device-side wavelet transforms (the HIP path behind the plugin classes)."""
import ctypes

import torch

from .. import _lib
from .wavelets import get_filters


def _layout_and_shape(x, channels_last):
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D batch, got shape {tuple(x.shape)}")
    if channels_last:
        B, H, W, C = x.shape
        return _lib.WV_LAYOUT_NHWC, B, C, H, W
    B, C, H, W = x.shape
    return _lib.WV_LAYOUT_NCHW, B, C, H, W


def _in_dtype(x):
    if x.dtype == torch.uint8:
        return _lib.WV_DT_U8
    if x.dtype == torch.float32:
        return _lib.WV_DT_F32
    raise TypeError(f"images must be uint8 (0..255) or float32 (already scaled), got {x.dtype}")


def swt2d(x, wavelet="haar", level=1, channels_last=False, out_dtype=torch.float32, out=None, band_major=False):
    """Batch of images on the GPU -> [B, C, 4, H, W] level-`level` sub-bands (cA, cH, cV, cD).

    Same numbers as stacking ``SWTTransform(level, wavelet)(img)`` of the reference
    (custom_transforms.py:145-166) over the batch: uint8 input is divided by 255 in fp32,
    float32 input is used as is.  ``channels_last`` = the batch is [B, H, W, C] (PIL layout).
    ``out`` = preallocated result buffer (reused across steps, e.g. when the transform runs on its own stream).
    ``band_major`` = write ``[4, B, C, H, W]`` instead: every band is one contiguous NCHW batch, so the models'
    band split (multi_dino_attention.py:818 ``permute(2,0,1,3,4).contiguous()``, :745 ``x[..., i, :, :]``) is a
    view and the sub-band tensor is never copied a second time.
    """
    lib = _lib.require_gpu()
    if not x.is_cuda:
        raise ValueError("swt2d: input must live on the GPU (swt2d_host is the host twin)")
    x = x.contiguous()
    layout, B, C, H, W = _layout_and_shape(x, channels_last)
    lo, hi = get_filters(wavelet)
    if out_dtype == torch.float32:
        odt = _lib.WV_DT_F32
    elif out_dtype == torch.bfloat16:
        odt = _lib.WV_DT_BF16
    else:
        raise TypeError("out_dtype must be float32 or bfloat16")
    shape = (4, B, C, H, W) if band_major else (B, C, 4, H, W)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=x.device)
    elif tuple(out.shape) != shape or out.dtype != out_dtype or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"swt2d: `out` must be a contiguous {list(shape)} tensor of out_dtype on the input's device")
    if B == 0:
        return out
    if H % (1 << level) or W % (1 << level):
        # PyWavelets' message for swt2 on a bad size
        raise ValueError(f"Length of data must be even along the transform axis / divisible by 2**level; "
                         f"got {H}x{W} at level {level}")
    ws_bytes = lib.wv_swt2d_workspace_bytes(B, C, H, W, level, len(lo))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    flo, fhi = _lib.host_floats(lo), _lib.host_floats(hi)
    with torch.cuda.device(x.device):
        for b0 in range(0, B, 65535):
            b1 = min(B, b0 + 65535)
            if band_major:
                rc = lib.wv_swt2d_forward_ex(_lib.ptr(x[b0:b1]), _in_dtype(x), layout, _lib.ptr(out[0, b0:b1]), odt,
                                             _lib.WV_BANDS_OUTER, B * C * H * W, b1 - b0, C, H, W, level, flo, fhi,
                                             len(lo), _lib.ptr(ws), ctypes.c_size_t(ws_bytes), _lib.stream_ptr())
                if rc == -95:     # WV_ENOTSUP: shape outside the sliding kernel -> reference layout, re-laid out once
                    inner = swt2d(x, wavelet, level, channels_last=channels_last, out_dtype=out_dtype)
                    out.copy_(inner.permute(2, 0, 1, 3, 4))
                    return out
                _lib.check(rc, "wv_swt2d_forward_ex")
            else:
                rc = lib.wv_swt2d_forward(_lib.ptr(x[b0:b1]), _in_dtype(x), layout, _lib.ptr(out[b0:b1]), odt,
                                          b1 - b0, C, H, W, level, flo, fhi, len(lo), _lib.ptr(ws),
                                          ctypes.c_size_t(ws_bytes), _lib.stream_ptr())
                _lib.check(rc, "wv_swt2d_forward")
    return out


def swt2d_place_output(x, wavelet="haar", level=1, channels_last=False, out_dtype=torch.float32, band_major=False,
                       candidates=8, launches=4, first=None):
    """A result buffer for swt2d(x, ..., out=buffer) placed where the kernel runs fastest, for callers that keep one buffer
    for many batches of x's shape (a serving loop, bench.py).  WHERE a multi-GB write target lies in HBM moves the kernel by
    about +-3 % (the first large allocation of a process up to +10 %; one allocation behaves the same at every offset inside
    it -- DESIGN.md 5), so `candidates` buffers are allocated side by side, the transform is timed on each (`launches`
    launches, HIP events on the current stream) and the fastest is returned; the others go back to the driver.
    `first`: a buffer of the right shape the caller already holds; it is candidate 0.
    -> (buffer, {"candidates", "probe_ms", "picked"})."""
    _lib.require_gpu()
    layout, B, C, H, W = _layout_and_shape(x.contiguous(), channels_last)
    shape = (4, B, C, H, W) if band_major else (B, C, 4, H, W)
    if first is not None and (tuple(first.shape) != shape or first.dtype != out_dtype or first.device != x.device):
        raise ValueError("swt2d_place_output: `first` does not have the output's shape / dtype / device")
    bufs, ms = [], []
    for c in range(max(1, int(candidates))):
        buf = first if (c == 0 and first is not None) else torch.empty(shape, dtype=out_dtype, device=x.device)
        bufs.append(buf)
        if candidates <= 1 or B == 0:
            break
        for _ in range(2):
            swt2d(x, wavelet, level, channels_last=channels_last, out_dtype=out_dtype, out=buf, band_major=band_major)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            swt2d(x, wavelet, level, channels_last=channels_last, out_dtype=out_dtype, out=buf, band_major=band_major)
        e1.record()
        torch.cuda.synchronize(x.device)
        ms.append(round(e0.elapsed_time(e1) / launches, 4))
    pick = ms.index(min(ms)) if ms else 0
    best = bufs[pick]
    del bufs, buf
    torch.cuda.empty_cache()
    return best, {"candidates": max(1, int(candidates)), "probe_ms": ms, "picked": pick}


def dwt2d(x, wavelet="haar", level=1, channels_last=False):
    """Batch of images on the GPU -> [B, C, 4, H', W'] coarsest-level bands of the decimated DWT
    (``pywt.wavedec2(..., mode='symmetric')``, DWTTransform at custom_transforms.py:197-201)."""
    lib = _lib.require_gpu()
    if not x.is_cuda:
        raise ValueError("dwt2d: input must live on the GPU (no CPU path in the product)")
    x = x.contiguous()
    layout, B, C, H, W = _layout_and_shape(x, channels_last)
    lo, hi = get_filters(wavelet)
    Hn, Wn = lib.wv_dwt_out_len(H, len(lo), level), lib.wv_dwt_out_len(W, len(lo), level)
    out = torch.empty((B, C, 4, Hn, Wn), dtype=torch.float32, device=x.device)
    if B == 0:
        return out
    ws_bytes = lib.wv_dwt2d_workspace_bytes(B, C, H, W, level, len(lo))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.wv_dwt2d_forward(_lib.ptr(x), _in_dtype(x), layout, _lib.ptr(out), B, C, H, W, level,
                                  _lib.host_floats(lo), _lib.host_floats(hi), len(lo), _lib.ptr(ws),
                                  ctypes.c_size_t(ws_bytes), _lib.stream_ptr())
        _lib.check(rc, "wv_dwt2d_forward")
    return out


def rawstack(x, copies=4, channels_last=False):
    """[B,C,H,W] -> [B,C,copies,H,W] identical planes (RawStackTransform, :172-188)."""
    lib = _lib.require_gpu()
    if not x.is_cuda:
        raise ValueError("rawstack: input must live on the GPU")
    x = x.contiguous()
    layout, B, C, H, W = _layout_and_shape(x, channels_last)
    out = torch.empty((B, C, copies, H, W), dtype=torch.float32, device=x.device)
    if B == 0:
        return out
    with torch.cuda.device(x.device):
        rc = lib.wv_rawstack_forward(_lib.ptr(x), _in_dtype(x), layout, _lib.ptr(out), _lib.WV_DT_F32,
                                     B, C, H, W, copies, _lib.stream_ptr())
        _lib.check(rc, "wv_rawstack_forward")
    return out


# ---------------------------------------------------------------------------------------------------------------
# Host twins (wv_*_forward_cpu): what the plugins' __call__ runs inside forked DataLoader workers, where the
# reference runs pywt (flikr_coco.py:59-60 -> custom_transforms.py:145-157) and the GPU is out of reach.
# Host tensors in, host tensors out; single-threaded native code, no HIP call.
def _host_args(x, channels_last, what):
    if x.is_cuda:
        raise ValueError(f"{what}: the host twin takes host tensors (use the device function for GPU tensors)")
    x = x.contiguous()
    layout, B, C, H, W = _layout_and_shape(x, channels_last)
    return x, layout, B, C, H, W


def swt2d_host(x, wavelet="haar", level=1, channels_last=False):
    """Host twin of swt2d: [B,C,H,W] (or [B,H,W,C]) uint8 / float32 on the CPU -> float32 [B,C,4,H,W]."""
    lib = _lib.load()
    x, layout, B, C, H, W = _host_args(x, channels_last, "swt2d_host")
    lo, hi = get_filters(wavelet)
    out = torch.empty((B, C, 4, H, W), dtype=torch.float32)
    if B == 0:
        return out
    if H % (1 << level) or W % (1 << level):
        raise ValueError(f"Length of data must be even along the transform axis / divisible by 2**level; "
                         f"got {H}x{W} at level {level}")
    rc = lib.wv_swt2d_forward_cpu(_lib.ptr(x), _in_dtype(x), layout, _lib.ptr(out), B, C, H, W, level,
                                  _lib.host_floats(lo), _lib.host_floats(hi), len(lo))
    _lib.check(rc, "wv_swt2d_forward_cpu")
    return out


def dwt2d_host(x, wavelet="haar", level=1, channels_last=False):
    """Host twin of dwt2d -> float32 [B,C,4,H',W']."""
    lib = _lib.load()
    x, layout, B, C, H, W = _host_args(x, channels_last, "dwt2d_host")
    lo, hi = get_filters(wavelet)
    Hn, Wn = lib.wv_dwt_out_len(H, len(lo), level), lib.wv_dwt_out_len(W, len(lo), level)
    out = torch.empty((B, C, 4, Hn, Wn), dtype=torch.float32)
    if B == 0:
        return out
    rc = lib.wv_dwt2d_forward_cpu(_lib.ptr(x), _in_dtype(x), layout, _lib.ptr(out), B, C, H, W, level,
                                  _lib.host_floats(lo), _lib.host_floats(hi), len(lo))
    _lib.check(rc, "wv_dwt2d_forward_cpu")
    return out


def rawstack_host(x, copies=4, channels_last=False):
    """Host twin of rawstack -> float32 [B,C,copies,H,W]."""
    lib = _lib.load()
    x, layout, B, C, H, W = _host_args(x, channels_last, "rawstack_host")
    out = torch.empty((B, C, copies, H, W), dtype=torch.float32)
    if B == 0:
        return out
    rc = lib.wv_rawstack_forward_cpu(_lib.ptr(x), _in_dtype(x), layout, _lib.ptr(out), B, C, H, W, copies)
    _lib.check(rc, "wv_rawstack_forward_cpu")
    return out

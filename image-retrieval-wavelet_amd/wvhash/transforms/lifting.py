"""Legacy lifting-scheme transforms with the reference's names and call conventions
(/root/reference/main/transforms/custom_transforms.py:14-117): ``HaarLifting``, ``Cdf97Lifting``, ``ResizeSubBands``,
``CustomTransform(decompose_levels, basis, coarse_only, ll_only)``.  The lifting arithmetic runs in libwvhash
(``wv_lifting2d_forward``, bit-identical to the reference's float32 torch ops); padding and the level cascade are host
logic.  Inputs are float tensors ``[C, H, W]`` or ``[N, C, H, W]`` (what ToTensor / Normalize hand over); a CPU tensor is
moved to the GPU for the call and the result comes back on its original device, so the classes also work as a
per-image transform in a ``num_workers=0`` pipeline.
"""
import ctypes
from collections.abc import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_BASIS = {"haar": 0, "cdf97": 1}


def lifting2d(x, basis):
    """One level on the GPU.  x: float32 CUDA [..., H, W] with H, W even -> (ll [..., H/2, W/2], hi [..., 3, H/2, W/2])."""
    lib = _lib.require_gpu()
    if not x.is_cuda:
        raise ValueError("lifting2d: input must live on the GPU (no CPU path in the product)")
    x = x.float().contiguous()
    *lead, H, W = x.shape
    planes = 1
    for d in lead:
        planes *= d
    ll = torch.empty((*lead, H // 2, W // 2), dtype=torch.float32, device=x.device)
    hi = torch.empty((*lead, 3, H // 2, W // 2), dtype=torch.float32, device=x.device)
    if planes == 0:
        return ll, hi
    ws_bytes = lib.wv_lifting2d_workspace_bytes(planes, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.wv_lifting2d_forward(_lib.ptr(x), planes, H, W, _BASIS[basis], _lib.ptr(ll), _lib.ptr(hi), _lib.ptr(ws),
                                      ctypes.c_size_t(ws_bytes), _lib.stream_ptr())
        _lib.check(rc, "wv_lifting2d_forward")
    return ll, hi


class _Lifting(nn.Module):
    basis = "haar"

    def __init__(self, n_levels=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.n_levels = n_levels

    def _pad(self, x):
        raise NotImplementedError

    def forward_one(self, x):
        return lifting2d(self._pad(x), self.basis)

    def forward(self, x):
        dev = x.device
        if not x.is_cuda:
            _lib.require_gpu()
            x = x.cuda()
        details, approx = [], []
        for _ in range(self.n_levels):
            x, high = self.forward_one(x)
            details.append(high.to(dev))
            approx.append(x.to(dev))
        return approx, details


class HaarLifting(_Lifting):
    basis = "haar"

    def _pad(self, x):
        h, w = x.shape[-2:]
        return F.pad(x, (0, w % 2, 0, h % 2))


class Cdf97Lifting(_Lifting):
    basis = "cdf97"

    def _pad(self, x):
        h, w = x.shape[-2:]
        return F.pad(x, (0, (4 - (w % 4)) % 4, 0, (4 - (h % 4)) % 4))


class ResizeSubBands(nn.Module):
    """Bilinear (antialiased) resize of a sub-band tensor: ``torchvision.transforms.functional.resize`` on tensors is
    ``torch.nn.functional.interpolate`` underneath, which is what runs here (torchvision is optional)."""

    def __init__(self, size, interpolation="bilinear", max_size=None, antialias=True):
        super().__init__()
        ok = isinstance(size, int) or (isinstance(size, Sequence) and len(size) in (1, 2)
                                       and all(isinstance(v, int) for v in size))
        if not ok:
            raise (TypeError if not isinstance(size, (int, Sequence)) else ValueError)(
                f"ResizeSubBands: size must be an int or a sequence of one or two ints, got {size!r}")
        self.size, self.max_size, self.antialias = size, max_size, antialias
        self.interpolation = getattr(interpolation, "value", interpolation)

    def forward(self, img):
        h, w = img.shape[-2:]
        size = self.size
        if isinstance(size, int) or len(size) == 1:            # shorter side -> size, aspect kept
            s = size if isinstance(size, int) else size[0]
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = s, int(s * long / short)
            if self.max_size is not None and new_long > self.max_size:
                new_short, new_long = int(self.max_size * new_short / new_long), self.max_size
            size = (new_long, new_short) if w <= h else (new_short, new_long)
        lead = img.shape[:-2]
        x = img.reshape(-1, 1, h, w).float()
        y = F.interpolate(x, size=tuple(size), mode=self.interpolation, align_corners=False if self.interpolation != "nearest" else None,
                          antialias=self.antialias and self.interpolation in ("bilinear", "bicubic"))
        return y.reshape(*lead, *size).to(img.dtype)


WAVELET_DICT = {"haar": HaarLifting, "cdf97": Cdf97Lifting}


class CustomTransform:
    """Tensor-side lifting DWT of the legacy CNN configs (custom_transforms.py:90-117).  Output by option pair:

    ==========  ===========  ==============================================================
    ll_only     coarse_only  result (band axis = dim -3)
    ==========  ===========  ==============================================================
    False       True         coarsest level: approximation stacked on its three detail bands
    False       False        every level's bands stacked (single-level decompositions only)
    True        True         coarsest approximation alone
    True        False        every level's approximation stacked (single level only)
    ==========  ===========  ==============================================================
    """

    def __init__(self, decompose_levels=3, basis="haar", coarse_only=True, ll_only=False, device='cuda', dwt_mode='dwt'):
        self.dwt = WAVELET_DICT[basis](n_levels=decompose_levels)
        self.decompose_levels = decompose_levels
        self.coarse_only, self.ll_only = coarse_only, ll_only

    def __call__(self, img):
        approx, details = self.dwt(img)
        if self.coarse_only:
            top = approx[-1]
            return top if self.ll_only else torch.cat([top.unsqueeze(-3), details[-1]], dim=-3)
        if self.decompose_levels > 1:      # levels have different sizes: nothing to stack (the reference raises as well)
            what = "approx" if self.ll_only else "subbands"
            raise NotImplementedError(f"Full {what} not implemented yet for decompose_levels > 1 ")
        bands = list(approx) if self.ll_only else [a.unsqueeze(-3) for a in approx] + list(details)
        return torch.cat(bands, dim=-3)

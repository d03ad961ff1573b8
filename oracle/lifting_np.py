"""oracle/lifting_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy (float32, IEEE, no FMA) restatement of the reference's legacy lifting transforms:
  * 1-D lifting steps   /root/reference/main/transforms/wavelets/haar.py:21-43, cdf_97.py:33-73
  * zero-padded shifts  wavelets/utils.py:401-460 (pos_shift_4d / neg_shift_4d, PAD_MODE 'constant')
  * 2-D op + band order haar.py:69-86, cdf_97.py:119-133, utils.py:376-392 (LL top-left, LH bottom-left, HL top-right, HH)
  * 2-D scales          utils.py:58-77 (COEFFS_SCALES_V = 6: LL * 1/sqrt(2)^2, LH * 1, HL * 1, HH * sqrt(2))
  * padding + cascade   custom_transforms.py:14-55 (HaarLifting pads to even, Cdf97Lifting to a multiple of 4)
  * CustomTransform     custom_transforms.py:90-117

PARITY STATUS: pinned by the reference itself -- tests/golden/lifting_golden.npz holds outputs of the reference's own
fast_haar_2d_op / fast_cdf97_2d_op (tests/golden/make_golden_lifting.py); this restatement reproduces them bit for bit.

Only tests/ may import this module.
"""
import numpy as np

F32 = np.float32
A1, A2, A3, A4, K97 = -1.58613432, -0.05298011854, 0.8829110762, 0.4435068522, 1.149604398
SCALES_2D = np.array([1 / np.sqrt(2) ** 2, 1, 1, np.sqrt(2)], dtype=np.float32)     # COEFFS_SCALES_2D_v6


def _shift_pos(a, axis):     # a[i] -> a[i+1], zero at the end
    out = np.zeros_like(a)
    src = [slice(None)] * a.ndim; dst = [slice(None)] * a.ndim
    src[axis] = slice(1, None); dst[axis] = slice(0, -1)
    out[tuple(dst)] = a[tuple(src)]
    return out


def _shift_neg(a, axis):     # a[i] -> a[i-1], zero at the start
    out = np.zeros_like(a)
    src = [slice(None)] * a.ndim; dst = [slice(None)] * a.ndim
    src[axis] = slice(0, -1); dst[axis] = slice(1, None)
    out[tuple(dst)] = a[tuple(src)]
    return out


def _split(x, axis):
    ev = [slice(None)] * x.ndim; od = [slice(None)] * x.ndim
    ev[axis] = slice(0, None, 2); od[axis] = slice(1, None, 2)
    return x[tuple(ev)].astype(F32), x[tuple(od)].astype(F32)


def haar_1d(x, axis):
    ev, od = _split(x, axis)
    od1 = od + F32(-1.0) * ev
    ev1 = ev + F32(0.5) * od1
    k = np.sqrt(2.0)
    return np.concatenate([F32(k) * ev1, F32(1.0 / k) * od1], axis=axis)


def cdf97_1d(x, axis):
    ev, od = _split(x, axis)
    a1, a2, a3, a4 = F32(A1), F32(A2), F32(A3), F32(A4)
    od1 = od + (a1 * ev + a1 * _shift_pos(ev, axis))
    ev1 = ev + (a2 * _shift_neg(od1, axis) + a2 * od1)
    od2 = od1 + (a3 * ev1 + a3 * _shift_pos(ev1, axis))
    ev2 = ev1 + (a4 * _shift_neg(od2, axis) + a4 * od2)
    return np.concatenate([F32(K97) * ev2, F32(1.0 / K97) * od2], axis=axis)


def lifting_2d(x, basis):
    """x [..., H, W] with H, W even -> (LL, hi[..., 3, H/2, W/2] = stack(LH, HL, HH))."""
    op = haar_1d if basis == "haar" else cdf97_1d
    y = op(op(np.asarray(x, dtype=F32), -2), -1)          # across rows (H axis) first, then across cols
    h, w = y.shape[-2] // 2, y.shape[-1] // 2
    ll = y[..., :h, :w] * SCALES_2D[0]
    lh = y[..., h:, :w] * SCALES_2D[1]
    hl = y[..., :h, w:] * SCALES_2D[2]
    hh = y[..., h:, w:] * SCALES_2D[3]
    return ll, np.stack([lh, hl, hh], axis=-3)


def pad_for(x, basis):
    h, w = x.shape[-2:]
    ph, pw = (h % 2, w % 2) if basis == "haar" else ((4 - h % 4) % 4, (4 - w % 4) % 4)
    return np.pad(x, [(0, 0)] * (x.ndim - 2) + [(0, ph), (0, pw)])


def lifting_levels(x, basis, levels):
    """HaarLifting / Cdf97Lifting.forward: (approx list, details list)."""
    approx, details, cur = [], [], np.asarray(x, dtype=F32)
    for _ in range(levels):
        cur, hi = lifting_2d(pad_for(cur, basis), basis)
        approx.append(cur)
        details.append(hi)
    return approx, details


def custom_transform(x, decompose_levels=3, basis="haar", coarse_only=True, ll_only=False):
    l, h = lifting_levels(x, basis, decompose_levels)
    n = decompose_levels
    if not ll_only:
        if coarse_only:
            return np.concatenate([np.expand_dims(l[n - 1], -3), h[n - 1]], axis=-3)
        if n > 1:
            raise NotImplementedError("Full subbands not implemented yet for decompose_levels > 1 ")
        return np.concatenate([np.expand_dims(li, -3) for li in l] + list(h), axis=-3)
    if coarse_only:
        return l[n - 1]
    if n > 1:
        raise NotImplementedError("Full approx not implemented yet for decompose_levels > 1 ")
    return np.concatenate(l, axis=-3)

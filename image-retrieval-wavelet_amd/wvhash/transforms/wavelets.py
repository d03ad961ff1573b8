"""Decomposition filter taps (PyWavelets convention: dec_lo, dec_hi, index 0 first).

The reference passes a wavelet *name* to ``pywt.swt2`` (custom_transforms.py:164); PyWavelets is
not a dependency here, so the taps of the names its configs and studies use are tabulated
(SURVEY.md Appendix A): haar/db1 (all ``config/transform/*_swt.yaml``), db2 (BASELINE config c1),
db4 and bior4.4 (``studies/mflickr_wavelet_type_ablation.yaml:59-60``).  Orthogonal families
satisfy dec_hi[k] = (-1)^(k+1) * dec_lo[L-1-k].  A ``(dec_lo, dec_hi)`` pair may be passed instead
of a name for anything else.

Beyond the reference's own names: the other Daubechies orders db3, db5 ... db10 are COMPUTED (extremal phase, by spectral
factorisation in double precision -- the construction PyWavelets' tables come from; the same routine reproduces the
tabulated db2 / db4 to 4e-13).  They agree with PyWavelets to that accuracy, not digit for digit.
"""
from math import comb

_S2 = 0.7071067811865476

_DEC_LO = {
    "haar": [_S2, _S2],
    "db2": [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    "db4": [-0.010597401784997278, 0.032883011666982945, 0.030841381835986965, -0.18703481171888114,
            -0.02798376941698385, 0.6308807679295904, 0.7148465705525415, 0.23037781330885523],
}

_BIOR44 = (
    [0.0, 0.03782845550726404, -0.023849465019556843, -0.11062440441843718, 0.37740285561283066,
     0.8526986790088938, 0.37740285561283066, -0.11062440441843718, -0.023849465019556843,
     0.03782845550726404],
    [0.0, -0.06453888262869706, 0.04068941760916406, 0.41809227322161724, -0.7884856164055829,
     0.41809227322161724, 0.04068941760916406, -0.06453888262869706, 0.0, 0.0],
)

_ALIASES = {"db1": "haar", "sym2": "db2"}


def _qmf(dec_lo):
    L = len(dec_lo)
    return [(-1.0) ** (k + 1) * dec_lo[L - 1 - k] for k in range(L)]


_COMPUTED = {}


def _daubechies(N):
    """dec_lo of the extremal-phase Daubechies wavelet with N vanishing moments (2N taps), PyWavelets' order."""
    if N not in _COMPUTED:
        import numpy as np
        P = [comb(N - 1 + k, k) for k in range(N)]                 # P(y), y = (2 - z - 1/z) / 4, ascending powers
        zs = []
        for y in np.roots(P[::-1]):
            r = np.roots([1.0, -(2.0 - 4.0 * y), 1.0])             # z + 1/z = 2 - 4y
            zs.append(r[np.argmin(np.abs(r))])                     # the root inside the unit circle: minimum phase
        h = np.poly(np.concatenate([[-1.0] * N, zs])).real         # (1 + 1/z)^N prod (1 - z_i / z)
        h = h / h.sum() * 2.0 ** 0.5
        _COMPUTED[N] = [float(v) for v in h[::-1]]
    return list(_COMPUTED[N])


_COMPUTED_NAMES = {f"db{n}": n for n in (3, 5, 6, 7, 8, 9, 10)}


def get_filters(wavelet):
    """-> (dec_lo, dec_hi) lists of python floats."""
    if not isinstance(wavelet, str):
        lo, hi = wavelet
        lo, hi = [float(v) for v in lo], [float(v) for v in hi]
        if len(lo) != len(hi) or len(lo) < 1:
            raise ValueError("custom wavelet: dec_lo and dec_hi must have the same non-zero length")
        return lo, hi
    name = _ALIASES.get(wavelet, wavelet)
    if name == "bior4.4":
        return list(_BIOR44[0]), list(_BIOR44[1])
    if name in _DEC_LO:
        lo = list(_DEC_LO[name])
        return lo, _qmf(lo)
    if name in _COMPUTED_NAMES:
        lo = _daubechies(_COMPUTED_NAMES[name])
        return lo, _qmf(lo)
    raise ValueError(f"Unknown wavelet name '{wavelet}', check wavelist() for the list of available "
                     f"builtin wavelets: {wavelist()}")


def wavelist():
    return sorted(list(_DEC_LO) + ["bior4.4"] + list(_ALIASES) + list(_COMPUTED_NAMES))

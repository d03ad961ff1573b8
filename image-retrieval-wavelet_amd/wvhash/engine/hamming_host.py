"""Host-side Hamming retrieval primitives: the `_cpu` twins of libwvhash.so (csrc/host_rank.cpp) behind the names of
engine/hamming.py, on CPU tensors.  Used ONLY by CustomCalculator(device='cpu') -- the configuration the reference itself
runs its calculator in (/root/reference/main/engine/evaluate.py:76-81, accuracy_calculator.py:290-293) and BASELINE config
c0 ("CPU ... plumbing, no GPU").  Same integers as the kernels; average precision bit-identical (same summation order).
No GPU is touched and none is needed."""
import ctypes

import torch

from .. import _lib

SHARD_ROWS_MAX = 1 << 62        # no windowed-kernel limit on the host


def _words(nbits):
    return (nbits + 63) // 64


def _host(t, dtype=None):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    t = t.detach().cpu()
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous()


def _pack(src, mode, check, what):
    lib = _lib.load()
    src = _host(src, torch.float32)
    if src.dim() != 2:
        raise ValueError(f"{what}: expected a 2-D tensor, got {tuple(src.shape)}")
    rows, nbits = src.shape
    out = torch.empty((rows, _words(nbits)), dtype=torch.int64)
    flag = ctypes.c_int32(0)
    if rows:
        _lib.check(lib.wv_pack_bits_cpu(_lib.ptr(src), src.stride(0), _lib.ptr(out), rows, nbits, mode,
                                        ctypes.byref(flag) if check else None), "wv_pack_bits_cpu")
    if check and flag.value:
        if mode == 0:
            raise ValueError(f"{what}: codes must be exactly +1/-1 to be bit-packed (found 0, NaN or another value, e.g. sign(0))")
        raise ValueError(f"{what}: labels must be non-negative multi-hot values to be bit-packed")
    return out


def pack_codes(codes, check=True):
    return _pack(codes, 0, check, "pack_codes")


def pack_labels(labels, check=True):
    if labels.dim() == 1:
        raise ValueError("pack_labels: 1-D class-id labels are compared with ==, not packed")
    return _pack(labels, 1, check, "pack_labels")


def bit_counts(packed, nbits):
    lib = _lib.load()
    packed = _host(packed)
    counts = torch.empty(nbits, dtype=torch.int32)
    _lib.check(lib.wv_bit_counts_cpu(_lib.ptr(packed), packed.shape[0], nbits, _lib.ptr(counts)), "wv_bit_counts_cpu")
    return counts


def hamming_dist(q_packed, db, nbits=None):
    lib = _lib.load()
    q_packed, db = _host(q_packed), _host(db)
    Q, words = q_packed.shape
    N = db.shape[0]
    if db.shape[1] != words:
        raise ValueError("hamming_dist: query and database code widths differ")
    nbits = words * 64 if nbits is None else nbits
    if nbits > 255 or _words(nbits) != words:
        raise ValueError(f"hamming_dist: uint8 distances need nbits <= 255 matching the packed width (got nbits={nbits}, {words} words)")
    buf = torch.empty((Q, N), dtype=torch.uint8)
    if Q and N:
        _lib.check(lib.wv_hamming_dist_cpu(_lib.ptr(q_packed), _lib.ptr(db), _lib.ptr(buf), N, Q, N, words), "wv_hamming_dist_cpu")
    return buf


def hamming_topk(q_packed, db, nbits, k, idx_offset=0, workspace=None, want_dist=True, want_cum=False):
    """k nearest database rows per query, ascending (distance, index) -> (idx int32 [Q,k], dist uint8 [Q,k] or None)."""
    if want_cum:
        raise NotImplementedError("hamming_topk on the host: histograms are a by-product of the sharded GPU search only")
    lib = _lib.load()
    q_packed, db = _host(q_packed), _host(db)
    Q, words = q_packed.shape
    N = db.shape[0]
    if db.shape[1] != words or words != _words(nbits):
        raise ValueError("hamming_topk: code widths do not match nbits")
    idx = torch.empty((Q, k), dtype=torch.int32)
    dist = torch.empty((Q, k), dtype=torch.uint8) if want_dist else None
    if Q:
        _lib.check(lib.wv_hamming_topk_cpu(_lib.ptr(q_packed), _lib.ptr(db), _lib.ptr(idx), _lib.ptr(dist), Q, N, nbits, k,
                                           idx_offset), "wv_hamming_topk_cpu")
    return idx, dist


def map_at_k(idx, qlab_packed, dblab_packed, k=None):
    lib = _lib.load()
    idx = _host(idx, torch.int32) if idx.stride(-1) != 1 or idx.dtype != torch.int32 or idx.is_cuda else idx
    qlab_packed, dblab_packed = _host(qlab_packed), _host(dblab_packed)
    Q, kfull = idx.shape
    k = kfull if k is None else int(k)
    if not 1 <= k <= kfull:
        raise ValueError(f"map_at_k: k={k} outside the lists' length {kfull}")
    lw = qlab_packed.shape[1]
    if dblab_packed.shape[1] != lw:
        raise ValueError("map_at_k: label widths differ")
    ap = torch.empty(Q, dtype=torch.float32)
    nrel = torch.empty(Q, dtype=torch.int32)
    if Q:
        _lib.check(lib.wv_map_at_k_cpu(_lib.ptr(idx), idx.stride(0), Q, k, _lib.ptr(qlab_packed), _lib.ptr(dblab_packed), lw,
                                       _lib.ptr(ap), _lib.ptr(nrel)), "wv_map_at_k_cpu")
    return ap, nrel


def hit_prefix(idx, qlab_packed, dblab_packed):
    lib = _lib.load()
    idx = _host(idx, torch.int32)
    qlab_packed, dblab_packed = _host(qlab_packed), _host(dblab_packed)
    Q, k = idx.shape
    lw = qlab_packed.shape[1]
    if dblab_packed.shape[1] != lw:
        raise ValueError("hit_prefix: label widths differ")
    hits = torch.empty((Q, k), dtype=torch.int32)
    if Q:
        _lib.check(lib.wv_hit_prefix_cpu(_lib.ptr(idx), Q, k, _lib.ptr(qlab_packed), _lib.ptr(dblab_packed), lw, _lib.ptr(hits)),
                   "wv_hit_prefix_cpu")
    return hits


class PreparedDB:
    """Nothing to prepare on the host: keeps the packed codes (the calculator's code path is shared with the GPU)."""

    def __init__(self, db_packed, nbits=None, _virtual=True):
        self.packed = _host(db_packed)
        self.N, self.words = self.packed.shape
        self.nbits = nbits if nbits is not None else self.words * 64
        self.parts = None


def hamming_map_at_k(*args, **kwargs):
    return None                   # ranking and AP are two calls on the host (same numbers)

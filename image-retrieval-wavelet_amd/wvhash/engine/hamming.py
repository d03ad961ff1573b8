"""Device-side Hamming retrieval primitives (thin wrappers over the C ABI, include/wvhash.h).

Codes are +-1 fp32 rows in the reference (torch.sign output, multi_dino_attention.py:833); here
they are packed 64 per int64 word once and stay on the GPU.
"""
import ctypes

import torch

from .. import _lib


def _words(nbits):
    return (nbits + 63) // 64


def _pack(src, mode, check, what):
    lib = _lib.require_gpu()
    if src.dim() != 2:
        raise ValueError(f"{what}: expected a 2-D tensor, got {tuple(src.shape)}")
    src = src.to(device=_device_of(src), dtype=torch.float32)
    if src.stride(1) != 1:
        src = src.contiguous()
    rows, nbits = src.shape
    out = torch.empty((rows, _words(nbits)), dtype=torch.int64, device=src.device)
    flag = torch.zeros(1, dtype=torch.int32, device=src.device) if check else None
    if rows:
        with torch.cuda.device(src.device):
            rc = lib.wv_pack_bits(_lib.ptr(src), src.stride(0), _lib.ptr(out), rows, nbits, mode,
                                  _lib.ptr(flag), _lib.stream_ptr())
            _lib.check(rc, "wv_pack_bits")
    if check and rows and int(flag.item()):
        if mode == 0:
            raise ValueError(f"{what}: codes must be exactly +1/-1 to be bit-packed (found 0, NaN or another "
                             "value, e.g. sign(0)); use the float path get_knn(..., distance_metric='cosine')")
        raise ValueError(f"{what}: labels must be non-negative multi-hot values to be bit-packed")
    return out


def _device_of(t):
    if t.is_cuda:
        return t.device
    _lib.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def pack_codes(codes, check=True):
    """[N, nbits] +-1 -> int64 [N, ceil(nbits/64)], bit j of word w = codes[:, 64w+j] > 0."""
    return _pack(codes, 0, check, "pack_codes")


def pack_labels(labels, check=True):
    """[N, Lc] multi-hot (>= 0) -> int64 [N, ceil(Lc/64)]."""
    if labels.dim() == 1:
        raise ValueError("pack_labels: 1-D class-id labels are compared with ==, not packed")
    return _pack(labels, 1, check, "pack_labels")


def bit_counts(packed, nbits):
    lib = _lib.require_gpu()
    counts = torch.empty(nbits, dtype=torch.int32, device=packed.device)
    with torch.cuda.device(packed.device):
        rc = lib.wv_bit_counts(_lib.ptr(packed), packed.shape[0], nbits, _lib.ptr(counts), _lib.stream_ptr())
        _lib.check(rc, "wv_bit_counts")
    return counts


SHARD_ROWS_MAX = 32768      # wv_hamming_hist / wv_hamming_topk_rows16 take shards up to this many rows
RANK_K_MAX = 32639          # longest list the windowed kernel builds (its 16-bit cells count list bytes: 2 (k + 128) < 65536)


class PreparedDB:
    """A packed database laid out once for the kernels (wv_db_prepare): what index.add() is to the
    reference's faiss path (get_knn.py:54), except that it is built once per database, not per call."""

    def __init__(self, db_packed, nbits=None, _virtual=True):
        lib = _lib.require_gpu()
        if db_packed.dim() != 2 or db_packed.dtype != torch.int64:
            raise ValueError("PreparedDB: expected packed int64 codes [N, words] (see pack_codes)")
        self.packed = db_packed.contiguous()
        self.N, self.words = self.packed.shape
        self.nbits = nbits if nbits is not None else self.words * 64
        nbytes = lib.wv_db_prepared_bytes(self.N, self.words)
        self.blob = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.packed.device)
        if self.N:
            with torch.cuda.device(self.packed.device):
                rc = lib.wv_db_prepare(_lib.ptr(self.packed), self.N, self.words, _lib.ptr(self.blob),
                                       ctypes.c_size_t(nbytes), _lib.stream_ptr())
                _lib.check(rc, "wv_db_prepare")
        # A database beyond the windowed ranking kernel's 32,768 rows is ranked as contiguous VIRTUAL shards on the one
        # GPU (the row-sharded search of wvhash/parallel.py without the collectives: histograms -> prefix length ->
        # 16-bit list prefixes -> merge): 117,218 x 128-bit codes, 5000 queries, k = 5000: 1.1 ms instead of 1.83 ms for
        # the first-generation kernel.
        self.parts, self.per = None, None
        if SHARD_ROWS_MAX < self.N <= 64 * SHARD_ROWS_MAX and self.words <= 2 and _virtual:
            g = -(-self.N // SHARD_ROWS_MAX)
            self.per = -(-self.N // g)
            self.parts = [PreparedDB(self.packed[lo:min(self.N, lo + self.per)], self.nbits, _virtual=False)
                          for lo in range(0, self.N, self.per)]

    @property
    def shape(self):
        return self.packed.shape

    @property
    def device(self):
        return self.packed.device


def hamming_dist(q_packed, db, nbits=None):
    """-> uint8 [Q, N] view (row pitch padded to 64 bytes so every row store is 16-B aligned).
    `db`: packed int64 codes [N, words] or a PreparedDB.  Distances are bytes: codes of more than 255 bits could
    reach 256 (complementary codes), which would wrap to 0 -- refused; `nbits` tells a 193..255-bit code
    (4 words, fine) from a 256-bit one."""
    lib = _lib.require_gpu()
    Q, words = q_packed.shape
    prepared = isinstance(db, PreparedDB)
    N, dwords = (db.N, db.words) if prepared else db.shape
    if dwords != words:
        raise ValueError("hamming_dist: query and database code widths differ")
    if nbits is None:
        nbits = db.nbits if prepared else words * 64
    if nbits > 255 or _words(nbits) != words:
        raise ValueError(f"hamming_dist: uint8 distances need nbits <= 255 matching the packed width "
                         f"(got nbits={nbits}, {words} words); pass nbits for 193..255-bit codes")
    ld = (N + 63) // 64 * 64
    buf = torch.empty((Q, ld), dtype=torch.uint8, device=q_packed.device)
    if Q and N:
        with torch.cuda.device(q_packed.device):
            if prepared:
                rc = lib.wv_hamming_dist_prepared(_lib.ptr(q_packed), _lib.ptr(db.blob), _lib.ptr(buf), ld, Q, N,
                                                  words, _lib.stream_ptr())
            else:
                rc = lib.wv_hamming_dist(_lib.ptr(q_packed), _lib.ptr(db), _lib.ptr(buf), ld, Q, N, words,
                                         _lib.stream_ptr())
            _lib.check(rc, "wv_hamming_dist")
    return buf[:, :N]


class TopkWorkspace:
    """Reusable scratch for hamming_topk (the transposed database image)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        return self.buf


def hamming_topk(q_packed, db, nbits, k, idx_offset=0, workspace=None, want_dist=True, want_cum=False):
    """k nearest database rows per query, ascending (distance, index).
    `db`: packed int64 codes [N, words] or a PreparedDB.
    -> (idx int32 [Q,k], dist uint8 [Q,k] or None)   [+ cum int32 [Q, nbits+2] when want_cum:
    cum[q, b] = number of rows with distance < b]."""
    lib = _lib.require_gpu()
    Q, words = q_packed.shape
    prepared = isinstance(db, PreparedDB)
    N, dwords = (db.N, db.words) if prepared else db.shape
    if dwords != words or words != _words(nbits):
        raise ValueError("hamming_topk: code widths do not match nbits")
    dev = q_packed.device
    if prepared and db.parts and not want_cum and nbits <= 128 and Q:
        return _virtual_shards_topk(q_packed, db, nbits, k, idx_offset, want_dist)
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    dist = torch.empty((Q, k), dtype=torch.uint8, device=dev) if want_dist else None
    if want_cum:
        cum = torch.empty((Q, nbits + 2), dtype=torch.int32, device=dev)
        db_packed = db.packed if prepared else db
        use_prep = prepared and words <= 2
        ws = None
        if not use_prep:
            ws_bytes = lib.wv_hamming_topk_workspace_bytes(Q, N, words, k)
            ws = (workspace or TopkWorkspace()).get(ws_bytes, dev)
        with torch.cuda.device(dev):
            rc = lib.wv_hamming_topk_ex(_lib.ptr(q_packed), _lib.ptr(db_packed), _lib.ptr(db.blob) if use_prep else None,
                                        _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(cum), Q, N, nbits, k, idx_offset,
                                        _lib.ptr(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0),
                                        _lib.stream_ptr())
            _lib.check(rc, "wv_hamming_topk_ex")
        return idx, dist, cum
    with torch.cuda.device(dev):
        if prepared and words <= 2:
            rc = lib.wv_hamming_topk_prepared(_lib.ptr(q_packed), _lib.ptr(db.blob), _lib.ptr(idx), _lib.ptr(dist),
                                              Q, N, nbits, k, idx_offset, _lib.stream_ptr())
        else:
            db_packed = db.packed if prepared else db
            ws_bytes = lib.wv_hamming_topk_workspace_bytes(Q, N, words, k)
            ws = (workspace or TopkWorkspace()).get(ws_bytes, dev)
            rc = lib.wv_hamming_topk(_lib.ptr(q_packed), _lib.ptr(db_packed), _lib.ptr(idx), _lib.ptr(dist), Q, N,
                                     nbits, k, idx_offset, _lib.ptr(ws), ctypes.c_size_t(ws.numel()),
                                     _lib.stream_ptr())
        _lib.check(rc, "wv_hamming_topk")
    return idx, dist


def _prefix_need(cums, k):
    """cums int32 [G, Q, nbits + 2] -> the longest list prefix any shard owes any query (host int): local rows with
    distance <= T, T = the query's global k-th distance."""
    G, Q, _ = cums.shape
    T = (cums.sum(0)[:, 1:] >= k).int().argmax(dim=1)
    return int(torch.gather(cums, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max().item())


def _virtual_shards_topk(q_packed, db, nbits, k, idx_offset, want_dist):
    cums = torch.stack([hamming_hist(q_packed, part, nbits) for part in db.parts])
    send = max(1, min(k, db.per, _prefix_need(cums, k)))        # one host read, as in the un-hinted sharded search
    Q = q_packed.shape[0]
    lists = torch.zeros((len(db.parts), Q, send), dtype=torch.int16, device=q_packed.device)
    for g, part in enumerate(db.parts):
        w = min(send, part.N)
        lists[g, :, :w] = hamming_topk_rows16(q_packed, part, nbits, w)
    idx, dist = topk_merge_cum(lists, cums, db.per, k, nbits)
    if idx_offset:
        idx += int(idx_offset)
    return idx, (dist if want_dist else None)


def _shard_db_args(db, words, nbits, what):
    prepared = isinstance(db, PreparedDB)
    N, dwords = (db.N, db.words) if prepared else db.shape
    if dwords != words or words != _words(nbits):
        raise ValueError(f"{what}: code widths do not match nbits")
    return prepared, N


def hamming_hist(q_packed, db, nbits, workspace=None):
    """Cumulative distance histogram of every query over the rows of `db` (no list is built):
    cum int32 [Q, nbits + 2], cum[q, b] = rows with distance < b.  First step of the sharded search."""
    lib = _lib.require_gpu()
    Q, words = q_packed.shape
    prepared, N = _shard_db_args(db, words, nbits, "hamming_hist")
    dev = q_packed.device
    cum = torch.empty((Q, nbits + 2), dtype=torch.int32, device=dev)
    ws, ws_bytes = None, 0
    if not prepared:
        ws_bytes = lib.wv_hamming_topk_workspace_bytes(Q, N, words, 1)
        ws = (workspace or TopkWorkspace()).get(ws_bytes, dev)
    with torch.cuda.device(dev):
        rc = lib.wv_hamming_hist(_lib.ptr(q_packed), None if prepared else _lib.ptr(db), _lib.ptr(db.blob) if prepared else None,
                                 _lib.ptr(cum), Q, N, nbits, _lib.ptr(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0),
                                 _lib.stream_ptr())
        _lib.check(rc, "wv_hamming_hist")
    return cum


def hamming_topk_rows16(q_packed, db, nbits, k, workspace=None):
    """The k nearest rows of `db` per query, ascending (distance, row), as 16-bit LOCAL row numbers (int16 storage of
    uint16 values [Q, k]): the wire format of the sharded search (topk_merge_cum)."""
    lib = _lib.require_gpu()
    Q, words = q_packed.shape
    prepared, N = _shard_db_args(db, words, nbits, "hamming_topk_rows16")
    dev = q_packed.device
    rows = torch.empty((Q, k), dtype=torch.int16, device=dev)
    ws = None
    if not prepared:
        ws = (workspace or TopkWorkspace()).get(lib.wv_hamming_topk_workspace_bytes(Q, N, words, k), dev)
    with torch.cuda.device(dev):
        rc = lib.wv_hamming_topk_rows16(_lib.ptr(q_packed), None if prepared else _lib.ptr(db),
                                        _lib.ptr(db.blob) if prepared else None, _lib.ptr(rows), Q, N, nbits, k, _lib.ptr(ws),
                                        ctypes.c_size_t(ws.numel() if ws is not None else 0), _lib.stream_ptr())
        _lib.check(rc, "wv_hamming_topk_rows16")
    return rows


def hamming_shard_prefix(q_packed, db, nbits, k, workspace=None):
    """hamming_topk_rows16 and hamming_hist in one pass -> (rows int16 [Q, k], cum int32 [Q, nbits + 2])."""
    lib = _lib.require_gpu()
    Q, words = q_packed.shape
    prepared, N = _shard_db_args(db, words, nbits, "hamming_shard_prefix")
    dev = q_packed.device
    rows = torch.empty((Q, k), dtype=torch.int16, device=dev)
    cum = torch.empty((Q, nbits + 2), dtype=torch.int32, device=dev)
    ws = None
    if not prepared:
        ws = (workspace or TopkWorkspace()).get(lib.wv_hamming_topk_workspace_bytes(Q, N, words, k), dev)
    with torch.cuda.device(dev):
        rc = lib.wv_hamming_shard_prefix(_lib.ptr(q_packed), None if prepared else _lib.ptr(db),
                                         _lib.ptr(db.blob) if prepared else None, _lib.ptr(rows), _lib.ptr(cum), Q, N, nbits, k,
                                         _lib.ptr(ws), ctypes.c_size_t(ws.numel() if ws is not None else 0), _lib.stream_ptr())
        _lib.check(rc, "wv_hamming_shard_prefix")
    return rows, cum


def topk_merge(idx_in, dist_in, k, nbits):
    """[G,Q,kin] per-shard lists (contiguous row shards in rank order) -> global [Q,k]."""
    lib = _lib.require_gpu()
    G, Q, kin = idx_in.shape
    idx_in, dist_in = idx_in.contiguous(), dist_in.contiguous()
    idx = torch.empty((Q, k), dtype=torch.int32, device=idx_in.device)
    dist = torch.empty((Q, k), dtype=torch.uint8, device=idx_in.device)
    with torch.cuda.device(idx_in.device):
        rc = lib.wv_topk_merge(_lib.ptr(idx_in), _lib.ptr(dist_in), G, Q, kin, _lib.ptr(idx), _lib.ptr(dist), k,
                               nbits, _lib.stream_ptr())
        _lib.check(rc, "wv_topk_merge")
    return idx, dist


def topk_merge_cum(idx_local, cum, shard_rows, k, nbits, need_out=None):
    """Compact merge: idx_local int16 storage of uint16 LOCAL row numbers [G,Q,kin], cum int32 [G,Q,nbits+2]
    (per-shard cumulative distance histograms) -> global (idx int32 [Q,k], dist uint8 [Q,k]).
    need_out: int32 [1] device tensor (zeroed by the caller) that receives the longest prefix any shard had to contribute
    for any of these queries -- the lists are exact iff it is <= kin."""
    lib = _lib.require_gpu()
    G, Q, kin = idx_local.shape
    if idx_local.dtype != torch.int16 or cum.dtype != torch.int32 or tuple(cum.shape) != (G, Q, nbits + 2):
        raise ValueError("topk_merge_cum: expected int16 [G,Q,kin] local indices and int32 [G,Q,nbits+2] histograms")
    idx_local, cum = idx_local.contiguous(), cum.contiguous()
    idx = torch.empty((Q, k), dtype=torch.int32, device=idx_local.device)
    dist = torch.empty((Q, k), dtype=torch.uint8, device=idx_local.device)
    with torch.cuda.device(idx_local.device):
        rc = lib.wv_topk_merge_cum_need(_lib.ptr(idx_local), _lib.ptr(cum), G, Q, kin, shard_rows, _lib.ptr(idx),
                                        _lib.ptr(dist), k, nbits, _lib.ptr(need_out) if need_out is not None else None,
                                        _lib.stream_ptr())
        _lib.check(rc, "wv_topk_merge_cum_need")
    return idx, dist


def rank_from_dist(dist_matrix, nbits, k):
    lib = _lib.require_gpu()
    Q, N = dist_matrix.shape
    if dist_matrix.stride(1) != 1:
        dist_matrix = dist_matrix.contiguous()
    idx = torch.empty((Q, k), dtype=torch.int32, device=dist_matrix.device)
    dist = torch.empty((Q, k), dtype=torch.uint8, device=dist_matrix.device)
    with torch.cuda.device(dist_matrix.device):
        rc = lib.wv_rank_from_dist(_lib.ptr(dist_matrix), dist_matrix.stride(0), Q, N, nbits, _lib.ptr(idx),
                                   _lib.ptr(dist), k, _lib.stream_ptr())
        _lib.check(rc, "wv_rank_from_dist")
    return idx, dist


def map_at_k(idx, qlab_packed, dblab_packed, k=None):
    """Average precision per query over (the first k entries of) its ranked list -> (ap float32 [Q], nrel int32 [Q])."""
    lib = _lib.require_gpu()
    Q, kfull = idx.shape
    k = kfull if k is None else int(k)
    if not 1 <= k <= kfull:
        raise ValueError(f"map_at_k: k={k} outside the lists' length {kfull}")
    lw = qlab_packed.shape[1]
    if dblab_packed.shape[1] != lw:
        raise ValueError("map_at_k: label widths differ")
    if idx.stride(1) != 1:
        idx = idx.contiguous()
    ap = torch.empty(Q, dtype=torch.float32, device=idx.device)
    nrel = torch.empty(Q, dtype=torch.int32, device=idx.device)
    if Q:
        with torch.cuda.device(idx.device):
            rc = lib.wv_map_at_k_ld(_lib.ptr(idx), idx.stride(0), Q, k, _lib.ptr(qlab_packed), _lib.ptr(dblab_packed),
                                    lw, _lib.ptr(ap), _lib.ptr(nrel), _lib.stream_ptr())
            _lib.check(rc, "wv_map_at_k_ld")
    return ap, nrel


class PreparedLabels:
    """The database rows' packed label words laid out for the fused ranking + AP kernel (wv_rank_labels_prepare): one
    or two 64-bit multi-hot words per row (up to 128 classes), databases of at most 32,768 rows.  `ok` is False for anything else -- the caller then
    ranks and evaluates in two steps."""

    def __init__(self, dblab_packed, _virtual=True):
        lib = _lib.require_gpu()
        if dblab_packed.dim() != 2 or dblab_packed.dtype != torch.int64:
            raise ValueError("PreparedLabels: expected packed int64 label words [N, words] (see pack_codes)")
        self.packed = dblab_packed.contiguous()
        self.N, self.words = self.packed.shape
        self.parts = None
        if SHARD_ROWS_MAX < self.N <= 64 * SHARD_ROWS_MAX and self.words <= 2 and _virtual:      # as PreparedDB: virtual shards
            g = -(-self.N // SHARD_ROWS_MAX)
            per = -(-self.N // g)
            self.parts = [PreparedLabels(self.packed[lo:min(self.N, lo + per)], _virtual=False) for lo in range(0, self.N, per)]
        nbytes = lib.wv_rank_labels_prepared_bytes(self.N, self.words) if self.words <= 2 and self.N and not self.parts else 0
        self.ok = nbytes > 0 or bool(self.parts)
        self.blob = None
        if nbytes:
            self.blob = torch.empty(nbytes, dtype=torch.uint8, device=self.packed.device)
            with torch.cuda.device(self.packed.device):
                rc = lib.wv_rank_labels_prepare(_lib.ptr(self.packed), self.N, self.words, _lib.ptr(self.blob), ctypes.c_size_t(nbytes),
                                                _lib.stream_ptr())
                _lib.check(rc, "wv_rank_labels_prepare")


def hamming_map_at_k(q_packed, db, labels, qlab_packed, nbits, k):
    """mAP@k ingredients straight from the codes -> (ap float32 [Q], nrel int32 [Q]), or None when the shape is outside
    the fused kernel (the caller then runs hamming_topk + map_at_k, which return exactly the same numbers).
    db: PreparedDB; labels: PreparedLabels of the same rows; qlab_packed: int64 [Q, 1 or 2]."""
    lib = _lib.require_gpu()
    if not isinstance(db, PreparedDB) or not isinstance(labels, PreparedLabels):
        raise TypeError("hamming_map_at_k: needs a PreparedDB and PreparedLabels")
    Q, words = q_packed.shape
    if words != db.words or labels.N != db.N:
        raise ValueError("hamming_map_at_k: query / database / label shapes disagree")
    if not labels.ok or qlab_packed.shape[1] != labels.words or nbits > 128 or not 1 <= k <= db.N:
        return None
    if db.parts or labels.parts:                         # more than 32,768 rows: virtual shards, relevance strings, one merge
        if not (db.parts and labels.parts) or len(db.parts) != len(labels.parts) or not Q:
            return None
        cums = torch.stack([hamming_hist(q_packed, part, nbits) for part in db.parts])
        send = max(1, min(k, db.per, _prefix_need(cums, k)))
        if send > RANK_K_MAX:
            return None
        wires = torch.zeros((len(db.parts), Q, relbits_wire_words(send, nbits)), dtype=torch.int64, device=q_packed.device)
        for g, (part, lab) in enumerate(zip(db.parts, labels.parts)):
            if hamming_shard_relbits(q_packed, part, lab, qlab_packed, nbits, min(send, part.N), wire=wires[g], kin=send) is None:
                return None
        return merge_relbits_map(wires, send, k, nbits)
    ap = torch.empty(Q, dtype=torch.float32, device=q_packed.device)
    nrel = torch.empty(Q, dtype=torch.int32, device=q_packed.device)
    if Q:
        with torch.cuda.device(q_packed.device):
            rc = lib.wv_hamming_map_at_k(_lib.ptr(q_packed.contiguous()), _lib.ptr(db.blob), _lib.ptr(labels.blob),
                                         _lib.ptr(qlab_packed.contiguous()), labels.words, Q, db.N, nbits, k, _lib.ptr(ap), _lib.ptr(nrel),
                                         _lib.stream_ptr())
            if rc == -95:      # WV_ENOTSUP
                return None
            _lib.check(rc, "wv_hamming_map_at_k")
    return ap, nrel


def relbits_wire_words(kin, nbits):
    """int64 words per (query, shard) row of the sharded-mAP wire buffer: [histogram: nbits + 2 int32, padded to 8 bytes |
    relevance string: ceil(kin / 64) uint64]."""
    return (nbits + 3) // 2 + (kin + 63) // 64


def hamming_shard_relbits(q_packed, db, labels, qlab_packed, nbits, k, wire=None, kin=None):
    """What a shard contributes to the sharded mAP: per query the relevance string of its k nearest rows (bit p = the p-th
    nearest row shares a label with the query) and its cumulative distance histogram, side by side in ONE int64 buffer
    [Q, relbits_wire_words(kin, nbits)] (kin >= k: the prefix length the ranks agreed on; a shard with fewer rows leaves the
    tail of the string zero) -- a single all_to_all moves both.  `wire`: preallocated (zeroed) buffer.
    -> wire, or None when the shape is outside the fused kernel."""
    lib = _lib.require_gpu()
    if not isinstance(db, PreparedDB) or not isinstance(labels, PreparedLabels):
        raise TypeError("hamming_shard_relbits: needs a PreparedDB and PreparedLabels")
    Q, words = q_packed.shape
    if words != db.words or labels.N != db.N:
        raise ValueError("hamming_shard_relbits: query / database / label shapes disagree")
    if not labels.ok or qlab_packed.shape[1] != labels.words or nbits > 128 or not 1 <= k <= db.N:
        return None
    kin = k if kin is None else kin
    dev = q_packed.device
    ld = relbits_wire_words(kin, nbits)
    if wire is None:
        wire = torch.zeros((Q, ld), dtype=torch.int64, device=dev)
    elif tuple(wire.shape) != (Q, ld) or wire.dtype != torch.int64 or not wire.is_contiguous():
        raise ValueError("hamming_shard_relbits: wire buffer of the wrong shape")
    if Q:
        hist_words = (nbits + 3) // 2
        with torch.cuda.device(dev):
            rc = lib.wv_hamming_shard_relbits(_lib.ptr(q_packed.contiguous()), _lib.ptr(db.blob), _lib.ptr(labels.blob),
                                              _lib.ptr(qlab_packed.contiguous()), labels.words, wire.data_ptr() + 8 * hist_words, ld,
                                              wire.data_ptr(), 2 * ld, Q, db.N, nbits, k, _lib.stream_ptr())
            if rc == -95:      # WV_ENOTSUP
                return None
            _lib.check(rc, "wv_hamming_shard_relbits")
    return wire


def merge_relbits_map(wire, kin, k, nbits, need_out=None):
    """wire int64 [G, Q, relbits_wire_words(kin, nbits)] -- the rows hamming_shard_relbits made on G contiguous row shards in
    rank order -> (ap float32 [Q], nrel int32 [Q]) of the merged top-k lists, equal to map_at_k of the merged lists.
    need_out as in topk_merge_cum."""
    lib = _lib.require_gpu()
    G, Q, ld = wire.shape
    if wire.dtype != torch.int64 or ld != relbits_wire_words(kin, nbits):
        raise ValueError("merge_relbits_map: expected the int64 [G, Q, relbits_wire_words(kin, nbits)] wire buffer")
    wire = wire.contiguous()
    ap = torch.empty(Q, dtype=torch.float32, device=wire.device)
    nrel = torch.empty(Q, dtype=torch.int32, device=wire.device)
    if Q:
        hist_words = (nbits + 3) // 2
        with torch.cuda.device(wire.device):
            rc = lib.wv_merge_relbits_map(wire.data_ptr() + 8 * hist_words, ld, wire.data_ptr(), 2 * ld, G, Q, kin, k, nbits,
                                          _lib.ptr(ap), _lib.ptr(nrel), _lib.ptr(need_out) if need_out is not None else None,
                                          _lib.stream_ptr())
            _lib.check(rc, "wv_merge_relbits_map")
    return ap, nrel


def wire_histograms(wire, nbits):
    """int32 [.., nbits + 2] view of the histograms inside a wire buffer (tests, diagnostics)."""
    return wire.view(torch.int32).reshape(*wire.shape[:-1], 2 * wire.shape[-1])[..., :nbits + 2]


def hit_prefix(idx, qlab_packed, dblab_packed):
    """hits[q, p] = number of relevant entries among idx[q, :p+1] (uint32 counts as int32 tensor [Q, k])."""
    lib = _lib.require_gpu()
    Q, k = idx.shape
    lw = qlab_packed.shape[1]
    if dblab_packed.shape[1] != lw:
        raise ValueError("hit_prefix: label widths differ")
    hits = torch.empty((Q, k), dtype=torch.int32, device=idx.device)
    if Q:
        with torch.cuda.device(idx.device):
            rc = lib.wv_hit_prefix(_lib.ptr(idx.contiguous()), Q, k, _lib.ptr(qlab_packed), _lib.ptr(dblab_packed),
                                   lw, _lib.ptr(hits), _lib.stream_ptr())
            _lib.check(rc, "wv_hit_prefix")
    return hits

"""k-NN of queries vs references on the GPU -- same signature and return order as
/root/reference/main/engine/get_knn.py:9-24.

* ``distance_metric in ("hamming", "cosine")``: scores = q @ r.T, top-k largest (get_knn.py:63-66).
  +-1 codes take the bit-packed XOR/popcount kernel; the returned "distances" are the same inner
  products the reference returns (nbits - 2 * hamming).  Anything else takes the fp32 kernel.
* otherwise: L2, top-k smallest.  The reference's two back-ends disagree on the returned VALUES (same order):
  ``with_faiss=True`` (its default) returns faiss IndexFlatL2's SQUARED distances (get_knn.py:38-39,55),
  ``with_faiss=False`` returns torch.cdist's true distances (:67-69).  ``with_faiss`` selects the same here.
Ties are returned in ascending reference index (the reference's order inside a tie is whatever
torch.topk / faiss produce: implementation-defined).  Both back-ends are the same HIP kernels.
"""
import ctypes
import logging

import torch

from .. import _lib
from . import hamming as H

LOGGER = logging.getLogger("RETRIEVAL")


def _to_gpu(x):
    if not torch.is_tensor(x):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        _lib.require_gpu()
        x = x.cuda(non_blocking=True)
    return x


def _is_pm1(x):
    return bool(((x == 1) | (x == -1)).all().item())


def knn_float(references, queries, num_k, metric):
    lib = _lib.require_gpu()
    q = queries.float().contiguous()
    r = references.float().contiguous()
    Q, D = q.shape
    N = r.shape[0]
    idx = torch.empty((Q, num_k), dtype=torch.int32, device=q.device)
    val = torch.empty((Q, num_k), dtype=torch.float32, device=q.device)
    ws_bytes = lib.wv_knn_float_workspace_bytes(Q, N, D, num_k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        rc = lib.wv_knn_float(_lib.ptr(q), _lib.ptr(r), Q, N, D, metric, num_k, _lib.ptr(idx), _lib.ptr(val),
                              _lib.ptr(ws), ctypes.c_size_t(ws_bytes), _lib.stream_ptr())
        _lib.check(rc, "wv_knn_float")
    return val, idx


def knn_float_host(references, queries, num_k, metric):
    """wv_knn_float's host twin (csrc/host_knn.cpp) on CPU tensors: the same indices and values as the GPU path, bit for
    bit (the matrix cores' fp32 accumulation is an fmaf chain, which the twin walks in the same order)."""
    lib = _lib.load()
    q = queries.detach().float().contiguous()
    r = references.detach().float().contiguous()
    if q.is_cuda or r.is_cuda:
        raise ValueError("knn_float_host takes host tensors")
    Q, D = q.shape
    N = r.shape[0]
    idx = torch.empty((Q, num_k), dtype=torch.int32)
    val = torch.empty((Q, num_k), dtype=torch.float32)
    rc = lib.wv_knn_float_cpu(_lib.ptr(q), _lib.ptr(r), Q, N, D, metric, num_k, _lib.ptr(idx), _lib.ptr(val))
    _lib.check(rc, "wv_knn_float_cpu")
    return val, idx


def rank_scores(scores, k, descending=False, sqrt=False):
    """The k best columns of every row of a dense fp32 score matrix [Q, N] -- smallest first (descending: largest first), ties
    by ascending column -- as (values [Q, k], columns int32 [Q, k]); sqrt: the root of the returned values.  The ranking stage
    of wv_knn_float (wv_rank_scores); CPU tensors take its host twin.  Used to merge per-shard k-NN lists."""
    s = scores.detach().float().contiguous()
    Q, N = s.shape
    flags = (_lib.WV_RANK_DESCENDING if descending else 0) | (_lib.WV_RANK_SQRT if sqrt else 0)
    idx = torch.empty((Q, k), dtype=torch.int32, device=s.device)
    val = torch.empty((Q, k), dtype=torch.float32, device=s.device)
    if not s.is_cuda:
        _lib.check(_lib.load().wv_rank_scores_cpu(_lib.ptr(s), Q, N, k, flags, _lib.ptr(idx), _lib.ptr(val)), "wv_rank_scores_cpu")
        return val, idx
    lib = _lib.require_gpu()
    ws_bytes = lib.wv_rank_scores_workspace_bytes(Q, N, k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        _lib.check(lib.wv_rank_scores(_lib.ptr(s), Q, N, k, flags, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(ws),
                                      ctypes.c_size_t(ws_bytes), _lib.stream_ptr()), "wv_rank_scores")
    return val, idx


def get_knn(references, queries, num_k, embeddings_come_from_same_source, with_faiss=True, distance_metric="l2"):
    num_k += embeddings_come_from_same_source

    LOGGER.info("running k-nn with k=%d" % num_k)
    LOGGER.info("embedding dimensionality is %d" % references.size(-1))
    LOGGER.info(f"distance metric: {distance_metric}")

    references, queries = _to_gpu(references), _to_gpu(queries)
    if num_k > references.shape[0]:
        raise RuntimeError(f"selected index k out of range (k={num_k}, references={references.shape[0]})")

    nbits = references.shape[1]
    if distance_metric == "hamming" and nbits <= 128 and _is_pm1(references) and _is_pm1(queries):
        idx, dist = H.hamming_topk(H.pack_codes(queries, check=False), H.pack_codes(references, check=False),
                                   nbits, num_k)
        distances = float(nbits) - 2.0 * dist.float()
        indices = idx.long()
    elif distance_metric in ["hamming", "cosine"]:
        distances, idx = knn_float(references, queries, num_k, _lib.WV_METRIC_IP)
        indices = idx.long()
    else:
        # IndexFlatL2.search returns squared L2 (get_knn.py:38-39,55), torch.cdist the root (:67-69); same neighbours
        distances, idx = knn_float(references, queries, num_k,
                                   _lib.WV_METRIC_L2_SQUARED if with_faiss else _lib.WV_METRIC_L2)
        indices = idx.long()

    if embeddings_come_from_same_source:
        return indices[:, 1:], distances[:, 1:]
    return indices, distances

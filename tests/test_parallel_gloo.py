"""world_size-2 (and 3) gloo runs of the database-sharded retrieval path (wvhash/parallel.py) on the CPU.

The exchange logic (all_gather of query codes, per-shard ranking, all_to_all of the lists, merge, ragged
last shard padding) is exercised for real; the two GPU kernels it calls are replaced in the worker
processes by the oracle (oracle/ranking.py) -- tests may use the oracle, the product never does.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unpack(packed, nbits):
    """int64 [N, words] -> +-1 float [N, nbits] (inverse of the kernels' bit layout)."""
    bits = ((packed.unsqueeze(-1) >> torch.arange(64)) & 1).reshape(packed.shape[0], -1)[:, :nbits]
    return bits.float() * 2 - 1


def _pack(codes):
    n, nbits = codes.shape
    words = (nbits + 63) // 64
    pad = torch.zeros(n, words * 64)
    pad[:, :nbits] = (codes > 0).float()
    w = pad.reshape(n, words, 64).long()
    shifts = torch.arange(64)
    lo = (w[..., :63] << shifts[:63]).sum(-1)
    return lo + torch.where(w[..., 63] > 0, torch.tensor(-2 ** 63), torch.tensor(0))


def _fake_topk(q_packed, db, nbits, k, idx_offset=0, workspace=None, want_dist=True, want_cum=False):
    from oracle import ranking
    q, r = _unpack(q_packed, nbits), _unpack(db, nbits)
    idx, d = ranking.hamming_topk_stable(q, r, k)
    if not want_cum:
        return (idx + idx_offset).int(), d.to(torch.uint8)
    dm = ranking.hamming_matrix_u8(q, r)
    cum = torch.stack([(dm < b).sum(1) for b in range(nbits + 2)], dim=1).int()   # cum[q, b] = #rows with dist < b
    return (idx + idx_offset).int(), d.to(torch.uint8), cum


def _fake_hist(q_packed, db, nbits, workspace=None):
    from oracle import ranking
    dm = ranking.hamming_matrix_u8(_unpack(q_packed, nbits), _unpack(db, nbits))
    return torch.stack([(dm < b).sum(1) for b in range(nbits + 2)], dim=1).int()


def _fake_rows16(q_packed, db, nbits, k, workspace=None):
    from oracle import ranking
    idx, _ = ranking.hamming_topk_stable(_unpack(q_packed, nbits), _unpack(db, nbits), k)
    return idx.to(torch.int16)                                     # local rows (< 32768 in these tests)


def _fake_merge(idx_in, dist_in, k, nbits):
    G, Q, kin = idx_in.shape
    key = dist_in.permute(1, 0, 2).reshape(Q, G * kin).long()      # (shard, position) order per query
    ids = idx_in.permute(1, 0, 2).reshape(Q, G * kin)
    order = torch.argsort(key, dim=1, stable=True)[:, :k]
    return torch.gather(ids, 1, order), torch.gather(key, 1, order).to(torch.uint8)


def _fake_shard_prefix(q_packed, db, nbits, k, workspace=None):
    return _fake_rows16(q_packed, db, nbits, k), _fake_hist(q_packed, db, nbits)


def _label_words(rows):
    """0/1 label rows [n, Lc <= 128] -> int64 words [n, ceil(Lc / 64)], bit c of word w = class 64 w + c (pack_labels' layout)."""
    return _pack(rows.float() * 2 - 1)


def _wire_words(kin, nbits):
    return (nbits + 3) // 2 + (kin + 63) // 64


def _fake_shard_relbits(q_packed, db, labels, qlab_packed, nbits, k, wire=None, kin=None):
    """db: packed codes of the shard; labels: its label words [n, lw]; qlab_packed: label words of the queries [Q, lw].
    Fills the wire rows [histogram as int32 pairs | relevance string] like the kernel does."""
    from oracle import ranking
    kin = k if kin is None else kin
    idx, _ = ranking.hamming_topk_stable(_unpack(q_packed, nbits), _unpack(db, nbits), k)
    rel = ((labels[idx] & qlab_packed.unsqueeze(1)) != 0).any(-1)             # [Q, k]: shares a class in any label word
    Q, W, hw = rel.shape[0], (kin + 63) // 64, (nbits + 3) // 2
    bits = torch.zeros((Q, W * 64), dtype=torch.long)
    bits[:, :k] = rel.long()
    w = bits.reshape(Q, W, 64)
    sh = torch.arange(64)
    words = (w[..., :63] << sh[:63]).sum(-1) + torch.where(w[..., 63] > 0, torch.tensor(-2 ** 63), torch.tensor(0))
    if wire is None:
        wire = torch.zeros((Q, hw + W), dtype=torch.int64)
    wire[:, hw:] = words
    hist = torch.zeros((Q, 2 * hw), dtype=torch.int32)
    hist[:, :nbits + 2] = _fake_hist(q_packed, db, nbits)
    wire[:, :hw] = hist.view(torch.int64)
    return wire


def _fake_merge_relbits(wire, kin, k, nbits, need_out=None):
    """Expand every shard's string to one 0/1 entry per list position, merge by (distance, shard, position), AP as the
    reference computes it (fp32 quotients, mean over the hits)."""
    G, Q, ld = wire.shape
    hw = (nbits + 3) // 2
    cum = wire[..., :hw].contiguous().view(torch.int32).reshape(G, Q, 2 * hw)[..., :nbits + 2].long()
    relbits = wire[..., hw:]
    W = relbits.shape[-1]
    if need_out is not None:
        T = (cum.sum(0)[:, 1:] >= k).int().argmax(dim=1)
        owed = torch.gather(cum, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max()
        need_out.copy_(torch.maximum(need_out, owed.reshape(1).int()))
    sh = torch.arange(64)
    bits = ((relbits.unsqueeze(-1) >> sh) & 1).reshape(G, Q, W * 64)[:, :, :kin]
    pos = torch.arange(kin).view(1, 1, kin)
    d = (cum[:, :, 1:nbits + 2].unsqueeze(-1) <= pos.unsqueeze(2)).sum(2)
    d = torch.where(pos < cum[:, :, nbits + 1:nbits + 2].clamp(max=kin), d, torch.full_like(d, nbits + 1))
    key = d.permute(1, 0, 2).reshape(Q, G * kin)
    rel = bits.permute(1, 0, 2).reshape(Q, G * kin)
    order = torch.argsort(key, dim=1, stable=True)[:, :k]
    rel = torch.gather(rel, 1, order).float()
    hits = rel.cumsum(1)
    quo = (hits / torch.arange(1, rel.shape[1] + 1).float()) * rel
    nrel = rel.sum(1)
    ap = torch.where(nrel > 0, quo.double().sum(1) / nrel.clamp(min=1).double(), torch.zeros(Q, dtype=torch.float64))
    return ap.float(), nrel.int()


def _fake_merge_cum(idx_local, cum, shard_rows, k, nbits, need_out=None):
    """Expand the compact form (16-bit local rows + per-shard cumulative histograms) and merge as above."""
    G, Q, kin = idx_local.shape
    if need_out is not None:       # longest prefix any shard owed: cum[g, q, T_q + 1], T_q = global k-th distance of query q
        T = (cum.sum(0)[:, 1:] >= k).int().argmax(dim=1)
        owed = torch.gather(cum, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max()
        need_out.copy_(torch.maximum(need_out, owed.reshape(1).int()))
    ids = (idx_local.long() & 0xffff) + (torch.arange(G) * shard_rows).view(G, 1, 1)
    pos = torch.arange(kin).view(1, 1, kin)
    # distance of position p of a sorted list = number of boundaries cum[1:] that are <= p
    d = (cum[:, :, 1:nbits + 2].unsqueeze(-1) <= pos.unsqueeze(2)).sum(2)
    d = torch.where(pos < cum[:, :, nbits + 1:nbits + 2].clamp(max=kin), d, torch.full_like(d, nbits + 1))
    return _fake_merge(ids.int(), d.to(torch.uint8), k, nbits)


def _worker(rank, world, port, cases, nbits, ql, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash import parallel, synth
    from wvhash.engine import hamming as H
    H.hamming_topk, H.topk_merge, H.topk_merge_cum = _fake_topk, _fake_merge, _fake_merge_cum   # CPU stand-ins for the kernels
    H.hamming_hist, H.hamming_topk_rows16, H.hamming_shard_prefix = _fake_hist, _fake_rows16, _fake_shard_prefix
    out = {}
    for n_db, k in cases:
        q_all, r = synth.random_codes(world * ql, n_db, nbits, seed=3)
        lo, hi, _ = parallel.shard_bounds(n_db, world, rank)
        for trim in (True, False):      # histogram-trimmed exchange and full-length exchange must agree
            idx, d = parallel.sharded_hamming_topk(_pack(q_all[rank * ql:(rank + 1) * ql]), _pack(r[lo:hi]), nbits,
                                                   k, n_db, trim=trim)
            out[(n_db, k, trim)] = (idx, d)
        # the one-step trimmed form (what shards beyond the two-step kernels' range take) gives the same lists
        keep, H.SHARD_ROWS_MAX = H.SHARD_ROWS_MAX, 0
        idx1, d1 = parallel.sharded_hamming_topk(_pack(q_all[rank * ql:(rank + 1) * ql]), _pack(r[lo:hi]), nbits, k, n_db)
        H.SHARD_ROWS_MAX = keep
        assert torch.equal(idx1, out[(n_db, k, True)][0]) and torch.equal(d1, out[(n_db, k, True)][1])
        # hinted exchange (no host read): a generous hint is exact and verifies, a hint of 1 entry is flagged
        shard, kin = _pack(r[lo:hi]), min(k, parallel.shard_bounds(n_db, world, rank)[2])
        qs = _pack(q_all[rank * ql:(rank + 1) * ql])
        idx_h, d_h, need = parallel.sharded_hamming_topk(qs, shard, nbits, k, n_db, send_hint=kin, return_need=True)
        assert torch.equal(idx_h, out[(n_db, k, True)][0]) and torch.equal(d_h, out[(n_db, k, True)][1])
        assert parallel.exchange_ok([need], kin, kin)
        if int(need.item()) > 1:
            *_, need1 = parallel.sharded_hamming_topk(qs, shard, nbits, k, n_db, send_hint=1, return_need=True)
            assert not parallel.exchange_ok([need1], 1, kin)
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _map_worker(rank, world, port, cases, nbits, ql, out_dir, lc=12, p=0.2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash import parallel, synth
    from wvhash.engine import hamming as H
    H.hamming_shard_relbits, H.merge_relbits_map, H.relbits_wire_words = _fake_shard_relbits, _fake_merge_relbits, _wire_words
    H.hamming_hist = _fake_hist
    out = {}
    for n_db, k in cases:
        q_all, r = synth.random_codes(world * ql, n_db, nbits, seed=3)
        ql_all, rl = synth.multi_hot_labels(world * ql, lc, p, 5), synth.multi_hot_labels(n_db, lc, p, 6)
        lo, hi, per = parallel.shard_bounds(n_db, world, rank)
        sl = slice(rank * ql, (rank + 1) * ql)
        for hint in (min(k, per), 1, None):            # None: the exchange is sized exactly first (histograms, two all-reduces)
            parallel.TRACE = parallel.ExchangeTrace()
            ap, nrel, need = parallel.sharded_hamming_map_at_k(_pack(q_all[sl]), _label_words(ql_all[sl]), _pack(r[lo:hi]),
                                                               _label_words(rl[lo:hi]), nbits, k, n_db, hint)
            calls, parallel.TRACE = parallel.TRACE.calls, None
            # a hinted (steady-state) call: ONE all_gather (codes + label words) and ONE all_to_all, nothing else
            assert calls == {"all_gather": 1, "all_to_all": 1, "all_reduce": 0 if hint is not None else 2}, calls
            out[(n_db, k, hint)] = (ap, nrel, need, max(1, min(min(k, per), hint)) if hint is not None else min(k, per))
    torch.save(out, os.path.join(out_dir, f"m{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cases,nbits,lc,p", [
    (2, [(1000, 300), (1001, 600), (64, 10)], 64, 12, 0.2), (3, [(500, 500), (77, 40)], 64, 12, 0.2),
    # the c3 shape class: 128-bit codes and COCO's 80 classes = TWO label words per row next to two code words
    (2, [(1000, 300), (1001, 600)], 128, 80, 0.036), (3, [(500, 500), (77, 40)], 128, 80, 0.036),
    # the world size of the driver's scaling run (8 ranks; ragged last shard): the exchange the first RCCL run will make
    (8, [(1001, 300)], 64, 38, 0.1)])
def test_sharded_map_exchange_of_relevance_strings(tmp_path, world, cases, nbits, lc, p):
    """The exchange behind sharded_hamming_map_at_k (codes + label words in one all_gather, relevance strings + histograms
    through one all_to_all, merge on the receiving rank) with CPU stand-ins for the two kernels: AP and hit counts of the
    unsharded oracle ranking whenever the reported need fits the prefix that was sent; a prefix of one entry is flagged.
    Reference: accuracy_calculator.py:203-231 over get_knn.py:41-44's row shards."""
    from oracle import ranking
    from wvhash import synth
    ql = 5
    port = 31500 + (os.getpid() + world * 11 + nbits) % 2000
    mp.spawn(_map_worker, args=(world, port, cases, nbits, ql, str(tmp_path), lc, p), nprocs=world, join=True)
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"m{rank}.pt"))
        for n_db, k in cases:
            q_all, r = synth.random_codes(world * ql, n_db, nbits, seed=3)
            ql_all, rl = synth.multi_hot_labels(world * ql, lc, p, 5), synth.multi_hot_labels(n_db, lc, p, 6)
            ref_idx, _ = ranking.hamming_topk_stable(q_all, r, k)
            sl = slice(rank * ql, (rank + 1) * ql)
            rel = ((rl[ref_idx[sl]] * ql_all[sl].unsqueeze(1)).sum(-1) > 0).float()
            hits = rel.cumsum(1)
            want = ((hits / torch.arange(1, k + 1).float()) * rel).double().sum(1) / rel.sum(1).clamp(min=1).double()
            per = (n_db + world - 1) // world
            for hint in (min(k, per), 1, None):
                ap, nrel, need, send = got[(n_db, k, hint)]
                if min(int(need.item()), min(k, per)) <= send:
                    assert torch.equal(nrel.long(), rel.sum(1).long()) and (ap.double() - want).abs().max() < 1e-6
                else:
                    assert hint == 1


# one process group per world size (spawning costs a torch import per rank), several shapes inside:
# even shards, a ragged last shard (padding path), k larger than a shard, tiny database
# nbits = 8: tie-heavy (9 distinct distances over hundreds of rows) -- a shard's count of rows at distance <= T exceeds the
# `kin` entries that were exchanged although the exchange was exact: exchange_ok must not flag it
@pytest.mark.parametrize("world,cases,nbits", [(2, [(1000, 300), (1001, 600), (64, 10)], 64), (3, [(500, 500), (77, 40)], 64),
                                               (2, [(1000, 40), (400, 350)], 8)])
def test_sharded_topk_equals_unsharded(tmp_path, world, cases, nbits):
    from oracle import ranking
    from wvhash import synth
    ql = 5
    port = 29500 + (os.getpid() + world * 7 + nbits) % 2000
    mp.spawn(_worker, args=(world, port, cases, nbits, ql, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"r{rank}.pt"))
        for n_db, k in cases:
            q_all, r = synth.random_codes(world * ql, n_db, nbits, seed=3)
            ref_idx, ref_d = ranking.hamming_topk_stable(q_all, r, k)
            sl = slice(rank * ql, (rank + 1) * ql)
            for trim in (True, False):
                assert torch.equal(got[(n_db, k, trim)][0].long(), ref_idx[sl]), (world, rank, n_db, k, trim)
                assert torch.equal(got[(n_db, k, trim)][1].long(), ref_d[sl])


def test_pack_unpack_helpers_match_oracle_layout():
    from wvhash import synth
    q, _ = synth.random_codes(7, 1, 128, seed=1)
    assert torch.equal(_unpack(_pack(q), 128), q)


def test_shard_bounds_cover_the_database():
    from wvhash.parallel import shard_bounds
    for n, w in [(25000, 8), (1001, 2), (5, 8), (117218, 8)]:
        spans = [shard_bounds(n, w, r)[:2] for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


# ----------------------------------------------------------------------------------------- real-valued k-NN, row-sharded
def _knn_worker(rank, world, port, n_db, D, ql, ks, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash import parallel, _lib
    g = torch.Generator().manual_seed(17)
    q_all, r = torch.randn(world * ql, D, generator=g), torch.randn(n_db, D, generator=g)
    dup = min(40, n_db // 2)
    r[n_db // 2:n_db // 2 + dup] = r[:dup].clone()                # duplicated rows: ties that straddle shards
    q_all[1] = r[3]                                               # an exact match (squared distance 0)
    lo, hi, _ = parallel.shard_bounds(n_db, world, rank)
    out = {}
    parallel.TRACE = parallel.ExchangeTrace()
    for k in ks:
        for metric in (_lib.WV_METRIC_IP, _lib.WV_METRIC_L2, _lib.WV_METRIC_L2_SQUARED):
            out[(k, metric)] = parallel.sharded_knn_float(q_all[rank * ql:(rank + 1) * ql], r[lo:hi], k, metric, n_db)
    calls = dict(parallel.TRACE.calls)
    parallel.TRACE = None
    assert calls == {"all_gather": 3 * len(ks), "all_to_all": 3 * len(ks), "all_reduce": 0}, calls
    torch.save(out, os.path.join(out_dir, f"knn{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_db", [(2, 1001), (3, 700), (3, 7)])
def test_sharded_float_knn_equals_unsharded(tmp_path, world, n_db):
    """wvhash.parallel.sharded_knn_float (faiss' sharded IndexFlatIP / IndexFlatL2, get_knn.py:35-52) over real gloo ranks
    with the library's host twins: one all_gather + one all_to_all per call, and the merged lists equal wv_knn_float_cpu on
    the whole database bit for bit -- k below and above a shard's rows, a ragged (or empty) last shard, ties across shards."""
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    from wvhash import _lib
    from wvhash.engine.get_knn import knn_float_host
    D, ql = 12, 4
    ks = [1, min(5, n_db), min(400, n_db), n_db]
    port = 33600 + (os.getpid() + world * 13 + n_db) % 2000
    mp.spawn(_knn_worker, args=(world, port, n_db, D, ql, ks, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(17)
    q_all, r = torch.randn(world * ql, D, generator=g), torch.randn(n_db, D, generator=g)
    dup = min(40, n_db // 2)
    r[n_db // 2:n_db // 2 + dup] = r[:dup].clone()
    q_all[1] = r[3]
    for rank in range(world):
        got = torch.load(os.path.join(tmp_path, f"knn{rank}.pt"))
        for k in ks:
            for metric in (_lib.WV_METRIC_IP, _lib.WV_METRIC_L2, _lib.WV_METRIC_L2_SQUARED):
                v0, i0 = knn_float_host(r, q_all[rank * ql:(rank + 1) * ql], k, metric)
                v, i = got[(k, metric)]
                assert torch.equal(i, i0), (rank, k, metric)
                assert torch.equal(v.view(torch.int32), v0.view(torch.int32)), (rank, k, metric)

"""Database-sharded Hamming retrieval across the GPUs of one node (one process per GPU, RCCL).

Reference analogue: faiss.index_cpu_to_all_gpus(index, co) with co.shards = True
(/root/reference/main/engine/get_knn.py:41-44): database rows split across GPUs, brute-force search
per shard, per-shard top-k merged on the HOST.  Here the shards stay resident in HBM as packed codes,
and the only exchange steps are
  1. all_gather of the packed query codes (Q * nbits/8 bytes per rank -- tiny), so every rank can rank
     every query against its shard;
  2. all_to_all of the per-shard list prefixes: rank r receives, from every shard, the lists of ITS queries only
     (an all_gather would move world_size times more) -- as 16-bit local row numbers plus the shard's cumulative
     distance histogram per query (compact form), or int32 index + uint8 distance when a shard has > 65536 rows;
  3. a local G-way merge on the GPU (wv_topk_merge_cum / wv_topk_merge), exact and identical for every world
     size because lists are ordered by (distance, global index) and shards are contiguous row ranges in rank order.
xGMI is point-to-point: with 8 GPUs fully connected both collectives are one direct exchange per peer.
"""
import torch
import torch.distributed as dist

from .engine import hamming as H


def shard_bounds(n_rows, world_size, rank):
    per = (n_rows + world_size - 1) // world_size
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi, per


def _cpu_staged(group):
    """gloo moves host memory only: stage through the CPU (rehearsal path; RCCL takes device tensors)."""
    return dist.get_backend(group) == "gloo"


class ExchangeTrace:
    """What the sharded search puts on the wire: every collective of this module is counted (calls, payload bytes of this
    rank) and, with timing=True, bracketed by HIP events on the calling stream (the collective's own stream is joined to it
    by torch before the call returns, so the pair spans queueing + transfer).  Install with `parallel.TRACE = ExchangeTrace()`;
    bench.py asserts one all_gather + one all_to_all per steady-state step with it and reports per-rank milliseconds."""

    def __init__(self, timing=False):
        self.timing = timing
        self.calls = {"all_gather": 0, "all_to_all": 0, "all_reduce": 0}
        self.bytes = {"all_gather": 0, "all_to_all": 0, "all_reduce": 0}
        self.events = []                                  # (name, start, end)

    def ms(self):
        out = {"all_gather": 0.0, "all_to_all": 0.0, "all_reduce": 0.0}
        for name, e0, e1 in self.events:
            out[name] += e0.elapsed_time(e1)
        return out


TRACE = None


class _traced:
    def __init__(self, name, t):
        self.name, self.t, self.e0 = name, t, None

    def __enter__(self):
        tr = TRACE
        if tr is not None:
            tr.calls[self.name] += 1
            tr.bytes[self.name] += self.t.numel() * self.t.element_size()
            if tr.timing and self.t.is_cuda:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and TRACE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            TRACE.events.append((self.name, self.e0, e1))
        return False


def _all_gather(out, inp, group):
    with _traced("all_gather", inp):
        if _cpu_staged(group) and inp.is_cuda:
            o, i = out.cpu(), inp.cpu()
            dist.all_gather_into_tensor(o, i, group=group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=group)


def _all_to_all(out, inp, group):
    with _traced("all_to_all", inp):
        if _cpu_staged(group) and inp.is_cuda:
            o, i = out.cpu(), inp.cpu()
            dist.all_to_all_single(o, i, group=group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, group=group)


def pad_value(nbits):
    """Distance assigned to padding entries of ragged shard lists: sorts after every real entry."""
    return nbits + 1


def _all_reduce(t, op, group):
    with _traced("all_reduce", t):
        if _cpu_staged(group) and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=op, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op, group=group)


def sharded_hamming_topk(q_local, db_shard, nbits, k, n_total, group=None, workspace=None, trim=True,
                         send_hint=None, return_need=False, want_dist=True):
    """q_local: packed codes of THIS rank's queries [Ql, words]; db_shard: this rank's rows
    [lo:hi] of the packed database (tensor or PreparedDB).  Returns the global (idx int32 [Ql,k], dist uint8
    [Ql,k]) of the local queries.  Every rank must call with the same Ql.

    trim=True: the per-shard ranking also returns each query's cumulative distance histogram; one all-reduce
    of it (Q*(nbits+2)*4 bytes) gives every rank the global k-th distance T of every query, a shard then only
    has to send its list prefix with distance <= T.  The prefix length used is the maximum over all queries
    and shards (one scalar MAX all-reduce + one host read), so the exchange stays a fixed-size all_to_all --
    typically ~k/world + ties entries per query instead of min(k, shard rows).

    send_hint (with trim): prefix length to exchange WITHOUT the host read -- for a steady stream of query batches
    (serving, bench.py) whose needed length is known from earlier batches.  The call then never synchronises with the
    host and runs no all-reduce (one ranking pass per shard, two all_to_alls, the merge); the result is exact iff the
    returned `need` (device int32 [1], return_need=True: the longest prefix any shard owed one of THIS rank's queries) is
    <= send_hint on every rank, which the caller checks whenever it next synchronises anyway (`exchange_ok`).

    want_dist=False: the caller only needs the ranked lists (mAP does: calculate_maphashing, accuracy_calculator.py:183-231,
    never looks at the distances once the order is known).  With one rank the distance row is then not written at all
    (41 instead of 50 us at 2048 x 25,000, k = 5000); the merge of the multi-rank path writes it either way."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = H.hamming_topk(q_local, db_shard, nbits, k, workspace=workspace, want_dist=want_dist)
        return (out[0], out[1], None) if return_need else out
    rank = dist.get_rank(group)
    Ql, words = q_local.shape
    lo, hi, per = shard_bounds(n_total, world, rank)
    kin = min(k, per)                                   # longest list a shard could have to send
    # 1. every rank needs every query
    q_all = torch.empty((world * Ql, words), dtype=q_local.dtype, device=q_local.device)
    _all_gather(q_all, q_local.contiguous(), group)
    # 2. rank all queries against the local shard
    n_local = hi - lo
    if not isinstance(db_shard, H.PreparedDB) and db_shard.shape[0] != n_local:
        raise ValueError(f"rank {rank}: shard has {db_shard.shape[0]} rows, expected rows [{lo}, {hi}) of {n_total}")
    k_local = min(kin, n_local)
    dev = q_local.device
    i = d = cum = None
    # Two-step form (shards the windowed kernel takes): histograms first -- cheap, no list -- so that every rank knows
    # each query's global k-th distance BEFORE any list is built; the shard then ranks only the prefix that can matter
    # (about k / world + ties entries instead of min(k, shard rows)) and writes it straight in the 16-bit wire format.
    two_step = trim and 0 < n_local <= H.SHARD_ROWS_MAX and per <= H.SHARD_ROWS_MAX
    if trim and send_hint is not None and per <= H.SHARD_ROWS_MAX:
        # Hinted steady state: the prefix length is known, so nothing has to be learned before the lists are built.
        # ONE ranking pass per shard (the `send` nearest rows as 16-bit local numbers + the complete histograms), the two
        # all_to_alls, the merge -- no all-reduce at all (3 collectives per step instead of 5).  Whether `send` was enough
        # is decided where the lists arrive: the merge kernel has every shard's histograms of its queries, derives each
        # query's global k-th distance T and reports the longest prefix any shard owed (`need`, device int32 [1]).
        send = max(1, min(kin, int(send_hint)))
        if n_local > 0:
            w = min(send, n_local)
            loc_s, cum = H.hamming_shard_prefix(q_all, db_shard, nbits, w, workspace=workspace)
            if w < send:
                loc_s = torch.nn.functional.pad(loc_s, (0, send - w))
        else:                                           # empty shard: contributes nothing
            loc_s = torch.zeros((world * Ql, send), dtype=torch.int16, device=dev)
            cum = torch.zeros((world * Ql, nbits + 2), dtype=torch.int32, device=dev)
        loc_r = torch.empty_like(loc_s)
        _all_to_all(loc_r.view(torch.uint8), loc_s.contiguous().view(torch.uint8), group)
        cum_r = torch.empty_like(cum)
        _all_to_all(cum_r, cum.contiguous(), group)
        need = torch.zeros(1, dtype=torch.int32, device=dev)
        out = H.topk_merge_cum(loc_r.view(world, Ql, send), cum_r.view(world, Ql, nbits + 2), per, k, nbits, need_out=need)
        return (out[0], out[1], need) if return_need else out
    if k_local > 0 and not two_step:
        if trim:
            # the compact exchange below ships histograms, not distance rows: do not even write them
            i, d, cum = H.hamming_topk(q_all, db_shard, nbits, k_local, idx_offset=lo, workspace=workspace,
                                       want_dist=per > 65536, want_cum=True)
        else:
            i, d = H.hamming_topk(q_all, db_shard, nbits, k_local, idx_offset=lo, workspace=workspace)
    if two_step:
        cum = H.hamming_hist(q_all, db_shard, nbits, workspace=workspace)
    send = kin
    need = None
    if trim:
        if cum is None:                                 # empty shard: contributes nothing
            cum = torch.zeros((world * Ql, nbits + 2), dtype=torch.int32, device=dev)
        cum_g = cum.clone()
        _all_reduce(cum_g, dist.ReduceOp.SUM, group)
        # T = first bin b with (#rows of the whole database with distance <= b) >= k
        T = (cum_g[:, 1:] >= k).int().argmax(dim=1)
        need = torch.gather(cum, 1, (T + 1).unsqueeze(1).long()).max().reshape(1)   # local rows with distance <= T
        if send_hint is None:
            _all_reduce(need, dist.ReduceOp.MAX, group)
            send = max(1, min(kin, int(need.item())))   # exact sizing: the one host read of the exchange
        else:
            # no host read and no MAX all-reduce: every rank sends `send_hint` entries, `need` stays THIS shard's own
            # requirement and the caller verifies need <= send_hint on every rank (exchange_ok + one flag all-reduce
            # whenever it synchronises anyway) -- one collective less in every steady-state step
            send = max(1, min(kin, int(send_hint)))
    if trim and per <= 65536:
        # compact exchange: 16-bit LOCAL row numbers (2 bytes/entry) and, instead of a distance row, the shard's
        # cumulative histogram of each query (a sorted list is fully described by it): 2 bytes per entry + 4*(nbits+2)
        # bytes per (query, shard) on the wire instead of 5 bytes per entry.  int16 storage, shipped as bytes (RCCL has
        # no 16-bit integer type).
        if two_step:
            w = min(send, n_local)
            loc_s = H.hamming_topk_rows16(q_all, db_shard, nbits, w, workspace=workspace)
            if w < send:                                # a shard shorter than the prefix: pad (never read by the merge)
                loc_s = torch.nn.functional.pad(loc_s, (0, send - w))
        else:
            loc_s = torch.zeros((world * Ql, send), dtype=torch.int16, device=dev)
            if k_local > 0:
                w = min(send, k_local)
                loc_s[:, :w] = (i[:, :w] - lo).to(torch.int16)   # wraps for rows >= 32768; the kernel reads uint16
        loc_r = torch.empty_like(loc_s)
        _all_to_all(loc_r.view(torch.uint8), loc_s.contiguous().view(torch.uint8), group)
        cum_r = torch.empty_like(cum)
        _all_to_all(cum_r, cum.contiguous(), group)
        # received layout: [shard g][my Ql queries][...]
        out = H.topk_merge_cum(loc_r.view(world, Ql, send), cum_r.view(world, Ql, nbits + 2), per, k, nbits)
        return (out[0], out[1], need) if return_need else out
    idx_s = torch.full((world * Ql, send), -1, dtype=torch.int32, device=dev)
    dist_s = torch.full((world * Ql, send), pad_value(nbits), dtype=torch.uint8, device=dev)
    if k_local > 0:
        w = min(send, k_local)
        idx_s[:, :w], dist_s[:, :w] = i[:, :w], d[:, :w]
    # 3. exchange: block j of my lists (queries of rank j) goes to rank j
    idx_r = torch.empty_like(idx_s)
    dist_r = torch.empty_like(dist_s)
    _all_to_all(idx_r, idx_s, group)
    _all_to_all(dist_r, dist_s, group)
    # received layout: [shard g][my Ql queries][send]  ->  merge
    out = H.topk_merge(idx_r.view(world, Ql, send), dist_r.view(world, Ql, send), k, nbits)
    return (out[0], out[1], need) if return_need else out


def sharded_hamming_map_at_k(q_local, qlab_local, db_shard, labels_shard, nbits, k, n_total, send_hint, group=None):
    """mAP@k ingredients of THIS rank's queries against the row-sharded database, without a single list on the wire:
    -> (ap float32 [Ql], nrel int32 [Ql], need int32 [1]) or None when the shape is outside the kernels (the caller then
    uses sharded_hamming_topk + map_at_k: same numbers).

    calculate_maphashing (accuracy_calculator.py:183-231) needs of a list entry only whether it is relevant.  So a shard
    ranks every query against its rows as before, but what it sends per query is the RELEVANCE STRING of its `send_hint`
    nearest rows (1 bit per entry; the shard knows its rows' labels, the queries' label words travel with their codes in
    the one all_gather) plus its cumulative histogram; the receiver interleaves the strings bin by bin and evaluates the
    merged string exactly as map_at_k evaluates a list.  Per step: one all_gather, one ranking pass per shard, ONE small
    all_to_all (histogram and string side by side: 264 + 136 bytes per query and shard at 8 GPUs instead of 264 + 2,128 in
    two), one merge kernel.
    q_local int64 [Ql, words]; qlab_local int64 [Ql, 1 or 2] (up to 128 classes: MIRFLICKR's 38 in one label word, COCO's 80
    in two); db_shard PreparedDB and labels_shard PreparedLabels of this rank's rows; send_hint: the prefix length (learn it
    from an earlier batch, check the returned `need` with exchange_ok), or None for a one-off call that sizes the exchange
    exactly first: a histogram pass per shard, one SUM and one MAX all-reduce and one host read (what evaluate_sharded
    uses) -- `need` is then <= the prefix that was sent by construction."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        out = H.hamming_map_at_k(q_local, db_shard, labels_shard, qlab_local, nbits, k)
        return None if out is None else (out[0], out[1], None)
    rank = dist.get_rank(group)
    Ql, words = q_local.shape
    lo, hi, per = shard_bounds(n_total, world, rank)
    n_local = hi - lo
    lwords = qlab_local.shape[1]                         # 1 label word (<= 64 classes) or 2 (COCO's 80, NUS-WIDE's 81)
    if per > H.SHARD_ROWS_MAX or min(k, per) > H.RANK_K_MAX or lwords not in (1, 2) or nbits > 128:
        return None                                      # decided from values every rank shares: no rank goes another way
    dev = q_local.device
    both = torch.cat([q_local, qlab_local], dim=1).contiguous()           # codes | label words: one collective
    both_all = torch.empty((world * Ql, words + lwords), dtype=both.dtype, device=dev)
    _all_gather(both_all, both, group)
    q_all, ql_all = both_all[:, :words].contiguous(), both_all[:, words:words + lwords].contiguous()
    kin = min(k, per)
    if send_hint is None:
        # exact sizing (one-off calls): every rank learns each query's global k-th distance T from the summed histograms, a
        # shard's prefix = its rows with distance <= T, the exchange length = the longest prefix anywhere
        cum = (H.hamming_hist(q_all, db_shard, nbits) if n_local > 0
               else torch.zeros((world * Ql, nbits + 2), dtype=torch.int32, device=dev))
        cum_g = cum.clone()
        _all_reduce(cum_g, dist.ReduceOp.SUM, group)
        T = (cum_g[:, 1:] >= k).int().argmax(dim=1)
        owed = torch.gather(cum, 1, (T + 1).unsqueeze(1).long()).max().reshape(1)
        _all_reduce(owed, dist.ReduceOp.MAX, group)
        send_hint = int(owed.item())
    send = max(1, min(kin, int(send_hint)))
    wire = torch.zeros((world * Ql, H.relbits_wire_words(send, nbits)), dtype=torch.int64, device=dev)
    if n_local > 0:
        got = H.hamming_shard_relbits(q_all, db_shard, labels_shard, ql_all, nbits, min(send, n_local), wire=wire, kin=send)
        if got is None:
            raise RuntimeError("sharded_hamming_map_at_k: this shard is outside the fused kernel although the shared checks passed")
        wire = got
    wire_r = torch.empty_like(wire)
    _all_to_all(wire_r, wire, group)                      # histogram + relevance string of a (query, shard) side by side
    need = torch.zeros(1, dtype=torch.int32, device=dev)
    ap, nrel = H.merge_relbits_map(wire_r.view(world, Ql, -1), send, k, nbits, need_out=need)
    return ap, nrel, need


def exchange_ok(needs, send_hint, kin):
    """True when every `need` a hinted call returned fits the prefix length that was exchanged (one host read for
    the whole list; call it where the host synchronises anyway)."""
    needs = [n for n in needs if n is not None]
    if not needs:
        return True
    # the merge kernels report max_g cum[g][T+1] unclamped; a shard never owes more than `kin` = min(k, shard rows) entries
    # (with many ties at the k-th distance the raw count exceeds it although `kin` entries were exchanged: exact)
    worst = min(int(torch.stack([n.reshape(()) for n in needs]).max().item()), int(kin))
    return worst <= max(1, min(int(kin), int(send_hint)))


def merge_knn_lists(vals, gidx, k, metric):
    """Merge of per-shard k-NN lists -- what faiss' sharded index does on the host (get_knn.py:41-44) -- on the device the
    lists live on.  vals float32 [G, Ql, kk]: every shard's list per query, in ITS ranking order, as inner products
    (WV_METRIC_IP) or SQUARED distances (both L2 metrics); gidx int32 [G, Ql, kk]: global row numbers; padding entries of a
    short shard carry the sentinel (-inf for IP, +inf for L2).  Laid side by side in shard order, candidates with equal values
    are in ascending global row order (a shard ranks ties by ascending row, shard s holds lower rows than shard s + 1), so
    ranking the candidate matrix with ties by ascending column (wv_rank_scores) gives exactly the unsharded list.
    -> (values [Ql, k], global rows int32 [Ql, k])."""
    from . import _lib
    from .engine.get_knn import rank_scores
    G, Ql, kk = vals.shape
    cand = vals.permute(1, 0, 2).reshape(Ql, G * kk).contiguous()
    rows = gidx.permute(1, 0, 2).reshape(Ql, G * kk)
    v, pos = rank_scores(cand, k, descending=metric == _lib.WV_METRIC_IP, sqrt=metric == _lib.WV_METRIC_L2)
    return v, torch.gather(rows, 1, pos.long())


def sharded_knn_float(q_local, db_shard, k, metric, n_total, group=None):
    """Real-valued k-NN over a row-sharded database, one process per GPU: the role faiss.index_cpu_to_all_gpus(shards=True)
    plays for IndexFlatIP / IndexFlatL2 in the reference (main/engine/get_knn.py:35-52).  q_local [Ql, D]: THIS rank's
    queries (every rank the same Ql); db_shard: rows [rank * per, ...) of the database, per = ceil(n_total / world).
    One all_gather of the queries, wv_knn_float on the shard for everybody's queries, one all_to_all of the per-shard lists
    (value bits and global row packed into 8 bytes per entry), merge_knn_lists.  Results are those of wv_knn_float on the
    whole database, bit for bit, for every world size.  Host tensors (gloo) take the library's host twins.
    -> (values [Ql, k], global rows int32 [Ql, k])."""
    from . import _lib
    from .engine.get_knn import knn_float, knn_float_host
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    if k > n_total:
        raise RuntimeError(f"selected index k out of range (k={k}, references={n_total})")
    q_local = q_local.float().contiguous()
    Ql, D = q_local.shape
    dev = q_local.device
    q_all = torch.empty((world * Ql, D), dtype=torch.float32, device=dev)
    if world > 1:
        _all_gather(q_all, q_local, group)
    else:
        q_all.copy_(q_local)
    kk = min(k, per)
    n_loc = db_shard.shape[0]
    shard_metric = _lib.WV_METRIC_IP if metric == _lib.WV_METRIC_IP else _lib.WV_METRIC_L2_SQUARED
    sentinel = float("-inf") if metric == _lib.WV_METRIC_IP else float("inf")
    vals = torch.full((world * Ql, kk), sentinel, dtype=torch.float32, device=dev)
    rows = torch.full((world * Ql, kk), -1, dtype=torch.int32, device=dev)
    k_loc = min(kk, n_loc)
    if k_loc:
        v, i = (knn_float if dev.type == "cuda" else knn_float_host)(db_shard, q_all, k_loc, shard_metric)
        vals[:, :k_loc] = v
        rows[:, :k_loc] = i + lo
    # one 8-byte word per entry: value bits | global row
    send = (vals.view(torch.int32).long() << 32) | (rows.long() & 0xFFFFFFFF)
    recv = torch.empty_like(send)
    if world > 1:
        _all_to_all(recv, send.contiguous(), group)      # [shard, Ql, kk] of this rank's queries
    else:
        recv = send
    recv = recv.view(world, Ql, kk)
    got_v = (recv >> 32).int().view(torch.float32)
    got_i = (recv & 0xFFFFFFFF).int()
    return merge_knn_lists(got_v, got_i, k, metric)

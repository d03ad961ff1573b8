"""wv_knn_float's one-kernel ranking (k_row_topk: value bins -> LDS list -> in-bin ranks) against the radix kernel it
replaces for k <= 15,360 and against the stable oracle.  Both rank the SAME fp32 score matrix (k_scores), so indices
and values must be equal bit for bit -- ascending (key, index), ties by ascending index (get_knn.py:60-71 semantics:
torch.topk's order inside a tie is implementation-defined; ours is the stable one)."""
import numpy as np
import pytest
import torch

from oracle import ranking
from wvhash import _lib
from wvhash.engine.get_knn import knn_float

pytestmark = pytest.mark.gpu

IP, L2 = _lib.WV_METRIC_IP, _lib.WV_METRIC_L2


def _pair(Q, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(Q, D, generator=g).cuda(), torch.randn(N, D, generator=g).cuda()


def _both(diag, r, q, k, metric):
    diag.delenv("WV_KNN_RADIX_ONLY", raising=False)
    v1, i1 = knn_float(r, q, k, metric)
    diag.setenv("WV_KNN_RADIX_ONLY", "1")
    v0, i0 = knn_float(r, q, k, metric)
    diag.delenv("WV_KNN_RADIX_ONLY")
    torch.cuda.synchronize()
    return (v1, i1), (v0, i0)


@pytest.mark.parametrize("Q,N,D,k,metric", [
    (64, 25000, 64, 5000, IP),        # the c1 shape
    (33, 25000, 32, 100, L2),
    (20, 5717, 16, 5717, IP),         # c0: k = N, every bin survives
    (7, 117224, 8, 5000, L2),         # c3 row length
    (9, 300, 8, 300, L2),             # shorter than the sample: range = exact min / max
    (5, 4097, 8, 1, IP),              # one more than the sample, k = 1
    (3, 70000, 4, 15360, L2),         # the largest k the LDS list takes
    (3, 70000, 4, 15361, L2),         # one more: radix kernels only
    (130, 1000, 12, 37, IP),
    (4, 2, 4, 2, L2), (4, 1, 4, 1, IP),
])
def test_row_topk_equals_the_radix_kernels(diag, Q, N, D, k, metric):
    q, r = _pair(Q, N, D, Q * 7 + N)
    (v1, i1), (v0, i0) = _both(diag, r, q, k, metric)
    assert torch.equal(i1, i0)
    assert torch.equal(v1.view(torch.int32), v0.view(torch.int32))


def test_rows_the_value_bins_cannot_take_are_handed_to_the_radix_kernels(diag):
    """One call, rows of every kind: smooth scores, a constant row, a row with one huge outlier in the sample (all other
    scores share a bin), a NaN row, integer-valued scores (ties en masse).  Equal to the radix kernels row by row."""
    Q, N, D, k = 40, 9000, 8, 1200
    q, r = _pair(Q, N, D, 5)
    r[0] = 3e4                               # |q . r0| ~ 1e5: stretches the sampled range of every row
    q[3] = 0.0                               # constant row (all scores 0)
    q[4] = float("nan")
    q[5:9] = torch.randint(-2, 3, (4, D), device="cuda").float()
    r[100:4000] = torch.randint(-2, 3, (3900, D), device="cuda").float()
    for metric in (IP, L2):
        (v1, i1), (v0, i0) = _both(diag, r, q, k, metric)
        assert torch.equal(i1, i0)
        assert torch.equal(v1.view(torch.int32), v0.view(torch.int32))


def test_forced_hand_over_of_every_row(diag):
    q, r = _pair(50, 6000, 16, 11)
    v1, i1 = knn_float(r, q, 700, IP)
    diag.setenv("WV_KNN_FORCE_TODO", "1")
    v2, i2 = knn_float(r, q, 700, IP)
    torch.cuda.synchronize()
    assert torch.equal(i1, i2) and torch.equal(v1, v2)


@pytest.mark.parametrize("metric,name", [(IP, "cosine"), (L2, "l2")])
def test_row_topk_against_the_stable_oracle(metric, name):
    """Release library (no switches): integer-valued embeddings make every score exact in fp32, so the oracle's stable
    ranking is the unique answer -- duplicated rows put ties on both sides of the k-th position."""
    g = torch.Generator().manual_seed(3)
    Q, N, D, k = 13, 6000, 8, 2500
    q = torch.randint(-3, 4, (Q, D), generator=g).float()
    r = torch.randint(-3, 4, (N, D), generator=g).float() + torch.arange(N).float()[:, None] % 7 * 0.125
    r[N // 2:] = r[: N - N // 2].clone()
    v, i = knn_float(r.cuda(), q.cuda(), k, metric)
    sd, si = ranking.knn_stable(r, q, k, name)
    assert torch.equal(i.cpu().long(), si.long())
    np.testing.assert_allclose(v.cpu().numpy(), sd.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("D", [1, 3, 4, 7, 12, 16, 20, 36, 52, 70, 100, 384])
def test_scores_are_exact_for_every_tail_of_the_k_loop(D):
    """k_scores walks k in chunks of 16 with a two-stage register pipeline; the last chunk may hold 4, 8 or 12 values.
    Integer-valued embeddings: every product and partial sum is exact in fp32, whatever the summation order, so the
    sorted scores must equal torch's exactly -- on edge tiles too (Q, N not multiples of 64)."""
    g = torch.Generator().manual_seed(D)
    Q, N = 130, 333
    q = torch.randint(-3, 4, (Q, D), generator=g).float()
    r = torch.randint(-3, 4, (N, D), generator=g).float()
    v, i = knn_float(r.cuda(), q.cuda(), N, IP)
    full = q @ r.t()
    assert torch.equal(torch.gather(full, 1, i.cpu().long()), v.cpu())
    assert torch.equal(v.cpu(), torch.sort(full, dim=1, descending=True).values)
    v2, i2 = knn_float(r.cuda(), q.cuda(), N, L2)
    d2 = ((q[:, None, :] - r[None, :, :]) ** 2).sum(-1)
    torch.testing.assert_close(v2.cpu(), torch.gather(d2, 1, i2.cpu().long()).sqrt(), rtol=3e-7, atol=0)   # sqrt: 1 ulp
    assert (v2[:, 1:] >= v2[:, :-1]).all()
    # metric 2 (faiss IndexFlatL2): the same neighbours, squared distances -- exact here, ties by ascending index
    v3, i3 = knn_float(r.cuda(), q.cuda(), N, _lib.WV_METRIC_L2_SQUARED)
    assert torch.equal(i3, i2)
    order = torch.argsort(d2, dim=1, stable=True)
    assert torch.equal(i3.cpu().long(), order) and torch.equal(v3.cpu(), torch.gather(d2, 1, order))


def test_query_chunks_of_a_long_database(diag):
    """3,000,000 rows: the 2 GiB score scratch holds 35 query rows, so 40 queries take two chunks (the fragment image of the
    queries is rebuilt per chunk, the hand-over flags are per chunk).  Integer-valued embeddings: exact scores, the stable
    order is the unique answer."""
    g = torch.Generator().manual_seed(1)
    Q, N, D, k = 40, 3_000_000, 4, 50
    q = torch.randint(-3, 4, (Q, D), generator=g).float().cuda()
    r = torch.randint(-3, 4, (N, D), generator=g).float().cuda()
    r[torch.randint(0, N, (200,), generator=g)] *= 5.0        # a few large scores so that the top k is not one tie bucket
    v, i = knn_float(r, q, k, IP)
    full = q @ r.t()
    order = torch.argsort(-full, dim=1, stable=True)[:, :k]
    assert torch.equal(i.long(), order)
    assert torch.equal(v, torch.gather(full, 1, order))
    del full, order
    # smooth scores (the value-bin kernel takes every row) against the radix kernel, same two chunks
    q, r = torch.randn(Q, D, generator=g).cuda(), torch.randn(N, D, generator=g).cuda()
    (v1, i1), (v0, i0) = _both(diag, r, q, k, L2)
    assert torch.equal(i1, i0) and torch.equal(v1.view(torch.int32), v0.view(torch.int32))


@pytest.mark.parametrize("D", [3, 4, 64, 70, 100, 384])
def test_host_twin_equals_the_kernels_bit_for_bit(D):
    """wv_knn_float_cpu (csrc/host_knn.cpp) on random real-valued embeddings: v_mfma_f32_32x32x2_f32 is an fmaf chain, the
    twin walks k in the order the kernel feeds it and forms the norms in the kernel's lane / butterfly order -- so indices
    AND values are equal bit for bit, for every metric (ties or not)."""
    from wvhash.engine.get_knn import knn_float_host
    g = torch.Generator().manual_seed(D)
    Q, N, k = 33, 3000, 200
    q, r = torch.randn(Q, D, generator=g), torch.randn(N, D, generator=g)
    r[5] = q[3]                                                     # an exact match: squared distance 0 (or the clamp)
    for metric in (IP, L2, _lib.WV_METRIC_L2_SQUARED):
        vg, ig = knn_float(r.cuda(), q.cuda(), k, metric)
        vh, ih = knn_float_host(r, q, k, metric)
        assert torch.equal(ig.cpu(), ih), metric
        assert torch.equal(vg.cpu().view(torch.int32), vh.view(torch.int32)), metric


@pytest.mark.parametrize("Q,N,k,flags", [(37, 5000, 300, 0), (5, 20000, 20000, _lib.WV_RANK_DESCENDING), (9, 70, 70, _lib.WV_RANK_SQRT),
                                         (3, 40000, 15360, _lib.WV_RANK_DESCENDING), (1, 1, 1, 0)])
def test_rank_scores_is_the_stable_order_of_a_given_matrix(Q, N, k, flags):
    """wv_rank_scores: the ranking stage of wv_knn_float on scores the caller made (the merge step of the sharded search).
    Quantised scores (many ties) against torch's stable sort; the host twin gives the same."""
    from wvhash.engine.get_knn import rank_scores
    g = torch.Generator().manual_seed(N + k)
    s = (torch.randn(Q, N, generator=g) * 4).round().abs() / 4 if flags & _lib.WV_RANK_SQRT else (torch.randn(Q, N, generator=g) * 8).round() / 8
    desc, root = bool(flags & _lib.WV_RANK_DESCENDING), bool(flags & _lib.WV_RANK_SQRT)
    v, i = rank_scores(s.cuda(), k, descending=desc, sqrt=root)
    order = torch.argsort(s, dim=1, descending=desc, stable=True)[:, :k]
    want = torch.gather(s, 1, order)
    assert torch.equal(i.cpu().long(), order)
    assert torch.equal(v.cpu(), want.sqrt() if root else want) or torch.allclose(v.cpu(), want.sqrt(), rtol=3e-7, atol=0)
    vh, ih = rank_scores(s, k, descending=desc, sqrt=root)
    assert torch.equal(ih, i.cpu()) and torch.equal(vh.view(torch.int32), v.cpu().view(torch.int32))


def test_shard_lists_merge_to_the_unsharded_search():
    """merge_knn_lists on one GPU: G = 2 ... 8 shards ranked one after the other, their lists laid side by side and ranked
    again -- the unsharded wv_knn_float lists bit for bit (ties across shards: duplicated rows; ragged last shard)."""
    from wvhash import parallel
    g = torch.Generator().manual_seed(8)
    Q, N, D, k = 21, 9001, 32, 1500
    q, r = torch.randn(Q, D, generator=g).cuda(), torch.randn(N, D, generator=g).cuda()
    r[N // 2:N // 2 + 64] = r[:64].clone()
    for metric in (IP, L2, _lib.WV_METRIC_L2_SQUARED):
        v0, i0 = knn_float(r, q, k, metric)
        for G in (2, 3, 8):
            per = (N + G - 1) // G
            kk = min(k, per)
            sm = IP if metric == IP else _lib.WV_METRIC_L2_SQUARED
            V = torch.full((G, Q, kk), float("-inf") if metric == IP else float("inf"), device="cuda")
            I = torch.full((G, Q, kk), -1, dtype=torch.int32, device="cuda")
            for s in range(G):
                sh = r[s * per:(s + 1) * per]
                kl = min(kk, sh.shape[0])
                v, i = knn_float(sh, q, kl, sm)
                V[s, :, :kl], I[s, :, :kl] = v, i + s * per
            v, i = parallel.merge_knn_lists(V, I, k, metric)
            assert torch.equal(i, i0) and torch.equal(v.view(torch.int32), v0.view(torch.int32)), (metric, G)

"""GPU parity of the HIP ranking path (through the C ABI) against oracle/ranking.py:
bit-exact distances, top-k indices (canonical tie-break) and sorted distance rows; AP within 1e-6
(fp32 quotient per hit like the reference, fp64 accumulation instead of torch.mean's fp32)."""
import numpy as np
import pytest
import torch

from oracle import ranking
from wvhash import synth
from wvhash.engine import CustomCalculator, get_knn, hamming as H

pytestmark = pytest.mark.gpu

AP_TOL = 1e-6


def t32(a):
    return torch.from_numpy(a.astype(np.float32))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/ranking_golden.npz")


def cases(g):
    return sorted({k.split("/")[0] for k in g.files if k.endswith("/topk_idx")})


def test_golden_cases_bit_exact(gold):
    for n in cases(gold):
        q, r = t32(gold[n + "/q"]).cuda(), t32(gold[n + "/r"]).cuda()
        ql, rl = t32(gold[n + "/ql"]).cuda(), t32(gold[n + "/rl"]).cuda()
        k, nbits = int(gold[n + "/k"][0]), q.shape[1]
        qp, rp = H.pack_codes(q), H.pack_codes(r)
        d = H.hamming_dist(qp, rp)
        np.testing.assert_array_equal(d.cpu().numpy(), gold[n + "/dist"].astype(np.uint8))
        idx, dk = H.hamming_topk(qp, rp, nbits, k)
        np.testing.assert_array_equal(idx.cpu().numpy(), gold[n + "/topk_idx"])
        np.testing.assert_array_equal(dk.cpu().numpy(), gold[n + "/topk_dist"])
        idx2, dk2 = H.rank_from_dist(d, nbits, k)
        assert torch.equal(idx2, idx) and torch.equal(dk2, dk)
        ap, nrel = H.map_at_k(idx, H.pack_labels(ql), H.pack_labels(rl))
        np.testing.assert_allclose(ap.cpu().numpy(), gold[n + "/ap_stable"], atol=AP_TOL)
        calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
        m = calc.calculate_maphashing(q, ql, r, rl, k)
        assert abs(m - gold[n + "/map_stable"][0]) < AP_TOL
        assert abs(calc.calculate_bit_balance(r) - gold[n + "/bit_balance"][0]) < 1e-6
        assert abs(calc.calculate_worst_bit_balance(r) - gold[n + "/bit_balance"][1]) < 1e-6
        np.testing.assert_array_equal(calc.calc_hamming_dist(q, r).cpu().numpy(), gold[n + "/dist"])


@pytest.mark.parametrize("Q,N,nbits,k", [(64, 5717, 16, 5717), (48, 25000, 64, 5000), (40, 19581, 64, 19581),
                                          (16, 30000, 128, 5000), (7, 1, 64, 1), (3, 257, 64, 200),
                                          (5, 4096, 32, 1), (9, 70001, 48, 3000)])
def test_seeded_sizes_against_oracle(Q, N, nbits, k):
    q, r = synth.random_codes(Q, N, nbits, seed=Q + N)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    d = H.hamming_dist(qp, rp)
    ref_d = ranking.hamming_matrix_u8(q, r)
    assert torch.equal(d.cpu().long(), ref_d)
    idx, dk = H.hamming_topk(qp, rp, nbits, k)
    ref_idx, ref_dk = ranking.hamming_topk_stable(q, r, k)
    assert torch.equal(idx.cpu().long(), ref_idx)
    assert torch.equal(dk.cpu().long(), ref_dk)
    prep = H.PreparedDB(rp, nbits)                      # prepared layouts: identical results
    assert torch.equal(H.hamming_dist(qp, prep), d)
    pi, pd = H.hamming_topk(qp, prep, nbits, k)
    assert torch.equal(pi, idx) and torch.equal(pd, dk)


def test_structured_codes_map_parity_c1_shape():
    """MIRFLICKR-like: N=25000, 64 bit, Lc=38, k=5000; oracle on 32 queries."""
    Q, N, nbits, k = 32, 25000, 64, 5000
    ql, rl = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
    m, ap = calc.calculate_maphashing(q, ql, r, rl, k, return_per_query=True)
    m_st, ap_st = ranking.calculate_maphashing(q, ql, r, rl, k, stable=True, return_per_query=True)
    np.testing.assert_allclose(ap.cpu().numpy(), ap_st, atol=AP_TOL)
    assert abs(m - m_st) < AP_TOL and m > 0.5
    # against the reference's literal (unstable) order: tie noise only (hard part 1)
    m_ref = ranking.calculate_maphashing(q, ql, r, rl, k, stable=False)
    assert abs(m - m_ref) < 2e-3


def test_edge_cases_constant_codes_zero_hits_topk_none():
    calc = CustomCalculator(k=10, distance_metric="hamming", with_faiss=False)
    ql, rl = synth.multi_hot_labels(6, 20, 0.07, 1), synth.multi_hot_labels(300, 20, 0.07, 2)
    ones_q, ones_r = torch.ones(6, 64), torch.ones(300, 64)
    assert abs(calc.calculate_maphashing(ones_q, ql, ones_r, rl, 50)
               - ranking.calculate_maphashing(ones_q, ql, ones_r, rl, 50, stable=True)) < AP_TOL
    zq = torch.zeros(3, 5); zq[:, 0] = 1
    zr = torch.zeros(40, 5); zr[:, 1] = 1
    q, r = synth.random_codes(3, 40, 16, 9)
    assert calc.calculate_maphashing(q, zq, r, zr, 10) == 0.0
    zr[7, 0] = 1
    a = calc.calculate_maphashing(q, zq, r, zr, None)
    b = calc.calculate_maphashing(q, zq, r, zr, 4000)          # topk > N clips like gnd[0:topk]
    c = ranking.calculate_maphashing(q, zq, r, zr, None, stable=True)
    assert abs(a - c) < AP_TOL and a == b
    m = calc.calculate_maphashing(q, zq, r, zr, "max_bin_count")
    assert abs(m - ranking.calculate_maphashing(q, zq, r, zr, "max_bin_count", stable=True)) < AP_TOL
    ids_q, ids_r = torch.tensor([1, 2, 3]), torch.arange(40) % 4      # 1-D class-id labels
    assert abs(calc.calculate_maphashing(q, ids_q, r, ids_r, 10)
               - ranking.calculate_maphashing(q, ids_q, r, ids_r, 10, stable=True)) < AP_TOL


def test_non_pm1_codes_are_rejected_not_mis_ranked():
    q, r = synth.random_codes(2, 50, 16, 1)
    r[3, 5] = 0.0     # sign(0)
    with pytest.raises(ValueError):
        H.pack_codes(r.cuda())
    with pytest.raises(ValueError):
        H.pack_labels(torch.tensor([[1.0, -1.0]]).cuda())
    lib_err = pytest.raises(ValueError)
    with lib_err:
        H.hamming_topk(H.pack_codes(q.cuda()), H.pack_codes(q.cuda()), 16, 3)   # k > N


def test_get_knn_hamming_matches_reference_scores(gold):
    n = "rand_q16_n500_b32"
    q, r = t32(gold[n + "/q"]), t32(gold[n + "/r"])
    idx, dist = get_knn(r, q, 50, False, with_faiss=True, distance_metric="hamming")
    assert idx.dtype == torch.int64 and dist.dtype == torch.float32 and tuple(idx.shape) == (16, 50)
    np.testing.assert_array_equal(dist.cpu().numpy(), gold[n + "/knn_ip"])      # same IP values
    sd, si = ranking.knn_stable(r, q, 50, "hamming")
    assert torch.equal(idx.cpu(), si)
    # reference (torch.topk) order: same index set in every complete tie bucket
    for i in range(16):
        assert ranking.bucket_sets(idx[i].cpu(), dist[i].cpu()) == \
            ranking.bucket_sets(torch.from_numpy(gold[n + "/knn_idx_ref"][i]).long(), t32(gold[n + "/knn_ip"][i]))
    idx2, dist2 = get_knn(r, r[:8], 5, True, distance_metric="hamming")          # same source
    si2 = ranking.knn_stable(r, r[:8], 6, "hamming")[1][:, 1:]
    assert torch.equal(idx2.cpu(), si2) and tuple(dist2.shape) == (8, 5)


def test_topk_merge_of_row_shards_equals_unsharded():
    Q, N, nbits, k = 24, 11000, 64, 3000
    q, r = synth.random_codes(Q, N, nbits, seed=5)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    full_idx, full_d = H.hamming_topk(qp, rp, nbits, k)
    for G in (2, 3, 7, 8):                               # 7: ragged last shard (padded lists)
        per = (N + G - 1) // G
        kin = min(k, per)
        idxs = torch.full((G, Q, kin), -1, dtype=torch.int32, device="cuda")
        ds = torch.full((G, Q, kin), nbits + 1, dtype=torch.uint8, device="cuda")   # padding of short shards
        for g in range(G):
            lo, hi = g * per, min(N, (g + 1) * per)
            kk = min(kin, hi - lo)
            i, d = H.hamming_topk(qp, rp[lo:hi].contiguous(), nbits, kk, idx_offset=lo)
            idxs[g, :, :kk], ds[g, :, :kk] = i, d
        mi, md = H.topk_merge(idxs, ds, k, nbits)
        assert torch.equal(mi, full_idx) and torch.equal(md, full_d)


@pytest.mark.parametrize("N,nbits,k,G,send", [(11000, 64, 3000, 8, None), (11000, 64, 3000, 7, 900), (70000, 128, 5000, 2, 4000),
                                              (999, 16, 999, 3, None), (40, 32, 7, 5, None)])
def test_compact_merge_from_histograms_equals_unsharded(N, nbits, k, G, send):
    """wv_topk_merge_cum: shards ship 16-bit local row numbers + their cumulative distance histograms (what the
    sharded search exchanges); the merged lists equal the unsharded ranking.  `send` < needed is never used by the
    search (it sizes the prefix from the global threshold), so a trimmed prefix must still give the exact top k
    whenever it covers every shard's entries below the global threshold -- here: send = None means full lists."""
    Q = 19
    q, r = synth.random_codes(Q, N, nbits, seed=7)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    full_idx, full_d = H.hamming_topk(qp, rp, nbits, k)
    per = (N + G - 1) // G
    kin = min(k, per)
    lists, cums = [], []
    for g in range(G):
        lo, hi = min(N, g * per), min(N, (g + 1) * per)
        kk = min(kin, hi - lo)
        loc = torch.zeros((Q, kin), dtype=torch.int16, device="cuda")
        cum = torch.zeros((Q, nbits + 2), dtype=torch.int32, device="cuda")
        if kk > 0:
            i, _, cum = H.hamming_topk(qp, rp[lo:hi].contiguous(), nbits, kk, idx_offset=lo, want_dist=False, want_cum=True)
            loc[:, :kk] = (i - lo).to(torch.int16)
        lists.append(loc)
        cums.append(cum)
    loc, cum = torch.stack(lists), torch.stack(cums)
    if send is not None:
        # the search's rule: prefix length = max over (query, shard) of the local rows with distance <= global T
        tot = cum.sum(0)
        T = (tot[:, 1:] >= k).int().argmax(dim=1)
        need = int(torch.gather(cum, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max().item())
        send = max(1, min(kin, need))
        loc = loc[:, :, :send].contiguous()
    mi, md = H.topk_merge_cum(loc, cum, per, k, nbits)
    assert torch.equal(mi, full_idx) and torch.equal(md, full_d)


def test_full_size_properties_c1():
    """Q=2048, N=25000, 64 bit, k=5000 (BASELINE c1): properties instead of a 51M-entry oracle."""
    Q, N, nbits, k = 2048, 25000, 64, 5000
    q, r = synth.random_codes(Q, N, nbits, seed=0)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    d = H.hamming_dist(qp, rp)
    idx, dk = H.hamming_topk(qp, rp, nbits, k)
    torch.cuda.synchronize()
    assert (dk[:, 1:] >= dk[:, :-1]).all()                                    # sorted
    assert torch.equal(torch.gather(d, 1, idx.long()), dk)                    # lists agree with matrix
    same = dk[:, 1:] == dk[:, :-1]
    assert (idx[:, 1:][same] > idx[:, :-1][same]).all()                       # stable inside ties
    srt = torch.sort(idx.long(), dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()                                  # no duplicates
    kth = dk[:, -1:].long()
    assert ((d.long() < kth).sum(1) <= k).all() and ((d.long() <= kth).sum(1) >= k).all()
    # checksum of checksums vs torch on the GPU's own matrix (independent of the ranking kernel)
    ref_sorted = torch.sort(d.long(), dim=1, stable=True).values[:, :k]
    assert torch.equal(ref_sorted, dk.long())
    # symmetry / identity of the distance kernel
    dq = H.hamming_dist(qp[:512], qp[:512])
    assert torch.equal(dq, dq.t()) and (dq.diagonal() == 0).all()
    # oracle on a slice
    ref_idx, ref_dk = ranking.hamming_topk_stable(q[:8], r, k)
    assert torch.equal(idx[:8].cpu().long(), ref_idx)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_float_knn_matches_reference_golden(gold, metric):
    """get_knn float branches (get_knn.py:63-69) on the committed reference vectors.  Values: fp32
    MFMA fmaf chain vs ATen's fp32 GEMM (cdist uses the |x|^2+|y|^2-2xy form above 25 rows):
    atol 2e-5; indices equal wherever neighbouring values differ by more than that."""
    q, r = torch.from_numpy(gold[f"float_{metric}/q"]), torch.from_numpy(gold[f"float_{metric}/r"])
    idx, dist = get_knn(r, q, 20, False, with_faiss=False, distance_metric=metric)
    ref_d, ref_i = gold[f"float_{metric}/dist"], gold[f"float_{metric}/idx"]
    np.testing.assert_allclose(dist.cpu().numpy(), ref_d, atol=2e-5)
    gaps = np.abs(np.diff(ref_d, axis=1)).min(axis=1) > 1e-4
    assert gaps.sum() >= 5
    np.testing.assert_array_equal(idx.cpu().numpy()[gaps], ref_i[gaps])


@pytest.mark.parametrize("metric,Q,N,D,k", [("l2", 33, 3000, 128, 100), ("cosine", 17, 5000, 384, 5000),
                                              ("hamming", 8, 777, 64, 50), ("l2", 5, 130, 12, 130)])
def test_float_knn_seeded_against_oracle(metric, Q, N, D, k):
    g = torch.Generator().manual_seed(Q * N)
    q, r = torch.randn(Q, D, generator=g), torch.randn(N, D, generator=g)
    if metric == "cosine":
        q, r = torch.nn.functional.normalize(q), torch.nn.functional.normalize(r)
    if metric == "hamming":                      # sign(0)-style codes: not +-1 -> float IP path
        q, r = torch.sign(q), torch.sign(r)
        r[0, 0] = 0.0
    idx, dist = get_knn(r, q, k, False, with_faiss=False, distance_metric=metric)
    sd, si = ranking.knn_stable(r, q, k, metric)
    np.testing.assert_allclose(dist.cpu().numpy(), sd.numpy(), atol=3e-5 if metric != "hamming" else 0)
    # every returned value is the true value of the returned index, and the list is sorted
    full = (q @ r.t()) if metric != "l2" else torch.cdist(q, r)
    np.testing.assert_allclose(torch.gather(full, 1, idx.cpu()).numpy(), dist.cpu().numpy(), atol=3e-5)
    d = dist.cpu()
    assert ((d[:, 1:] <= d[:, :-1] + 1e-6) if metric != "l2" else (d[:, 1:] >= d[:, :-1] - 1e-6)).all()
    if metric == "hamming":                      # integer scores: exact, ties by ascending index
        assert torch.equal(idx.cpu(), si)


def test_get_accuracy_driver_metrics():
    Q, N, nbits, k = 40, 3000, 64, 500
    ql, rl = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    from wvhash.engine import get_accuracy_calculator
    calc = get_accuracy_calculator(k=k, distance_metric="hamming", with_faiss=False, pr_rc_path=None,
                                   exclude=["mean_reciprocal_rank", "precision_at_1", "r_precision"])
    acc = calc.get_accuracy(q, ql, r, rl, False)
    assert set(acc) == {"maphashing", "map", "bit_balance", "worst_bit_balance", "rpr", "pr", "pr_rc"}
    m_st = ranking.calculate_maphashing(q, ql, r, rl, k, stable=True)
    assert abs(acc["maphashing"] - m_st) < AP_TOL
    # map_level0 = RetrievalMAP over the k-NN lists = same lists here -> same APs, mean over non-lone queries
    sd, si = ranking.knn_stable(r, q, k, "hamming")
    rel = torch.stack([ranking.label_comparison_fn(ql[i:i + 1], rl[si[i]])[0] for i in range(Q)])
    assert abs(acc["map"] - ranking.retrieval_map(sd, rel)) < AP_TOL
    assert abs(acc["bit_balance"] - ranking.calculate_bit_balance(r)) < 1e-6
    # secondary diagnostics (torchmetrics RetrievalRPrecision / RetrievalPrecision(top_k=1) / PR curve)
    assert abs(acc["rpr"] - ranking.retrieval_rprecision(rel)) < 1e-9
    assert abs(acc["pr"] - ranking.retrieval_precision_at_1(rel)) < 1e-9
    assert acc["pr_rc"] == 0
    pr_ref, rc_ref = ranking.retrieval_pr_curve(rel)
    pr, rc = calc.last_pr_rc
    assert torch.allclose(pr.cpu(), pr_ref, atol=1e-12) and torch.allclose(rc.cpu(), rc_ref, atol=1e-12)
    idx, acc2 = calc.get_accuracy(q, ql, r, rl, False, return_indices=True)
    assert torch.equal(idx.cpu(), si) and acc2 == acc


def test_pr_rc_hashing_full_gallery_curves(tmp_path):
    Q, N, nbits = 24, 1500, 32
    ql, rl = synth.multi_hot_labels(Q, 20, 0.07, 5), synth.multi_hot_labels(N, 20, 0.07, 6)
    ql[3] = 0                                    # a query without any relevant item: excluded from the mean
    q, r = synth.structured_codes(ql, nbits, 3, 7), synth.structured_codes(rl, nbits, 3, 8)
    from wvhash.engine import CustomCalculator
    out = tmp_path / "pr_rc.csv"
    calc = CustomCalculator(include=("pr_rc_hashing",), k=N, distance_metric="hamming", pr_rc_path=str(out))
    acc = calc.get_accuracy(q, ql, r, rl, False)
    assert acc == {"pr_rc_hashing": 0}
    lone = (ranking.label_comparison_fn(ql, rl).sum(1) > 0)
    pr_ref, rc_ref = ranking.pr_rc_hashing(q, ql, r, rl, lone, stable=True)
    pr, rc = calc.last_pr_rc
    assert torch.allclose(pr.cpu().float(), pr_ref, atol=1e-6) and torch.allclose(rc.cpu().float(), rc_ref, atol=1e-6)
    import pandas as pd
    df = pd.read_csv(out)
    assert list(df.columns) == ["pr", "rc"] and len(df) == N and abs(df["rc"].iloc[-1] - 1.0) < 1e-12


def test_coco_shape_128bit_u32_path_k5000_and_all():
    """BASELINE c3 shape: N = 117,218 codes of 128 bit (counters no longer fit 16 bits, columns longer than the
    register cache), k = 5000 and k = N (mAP@ALL); oracle on 4 queries."""
    Q, N, nbits = 4, 117218, 128
    ql, rl = synth.multi_hot_labels(Q, 80, 0.036, 1), synth.multi_hot_labels(N, 80, 0.036, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    prep = H.PreparedDB(rp, nbits)
    for k in (5000, N):
        idx, dk = H.hamming_topk(qp, prep, nbits, k)
        ref_idx, ref_dk = ranking.hamming_topk_stable(q, r, k)
        assert torch.equal(idx.cpu().long(), ref_idx) and torch.equal(dk.cpu().long(), ref_dk)
        calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
        m = calc.calculate_maphashing(q, ql, r, rl, k)
        assert abs(m - ranking.calculate_maphashing(q, ql, r, rl, k, stable=True)) < AP_TOL
    d = H.hamming_dist(qp, prep)
    assert torch.equal(d.cpu().long(), ranking.hamming_matrix_u8(q, r))


def test_voc_shape_16bit_full_gallery():
    """BASELINE c0 shape: N = 5717, 16 bit, k = N, Lc = 20."""
    Q, N, nbits = 96, 5717, 16
    ql, rl = synth.multi_hot_labels(Q, 20, 0.07, 1), synth.multi_hot_labels(N, 20, 0.07, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    calc = CustomCalculator(k=N, distance_metric="hamming", with_faiss=False)
    m, ap = calc.calculate_maphashing(q, ql, r, rl, N, return_per_query=True)
    m_ref, ap_ref = ranking.calculate_maphashing(q, ql, r, rl, N, stable=True, return_per_query=True)
    np.testing.assert_allclose(ap.cpu().numpy(), ap_ref, atol=AP_TOL)
    assert abs(m - m_ref) < AP_TOL


def test_cumulative_histogram_output():
    Q, N, nbits, k = 9, 7001, 64, 500
    q, r = synth.random_codes(Q, N, nbits, seed=11)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    for db in (rp, H.PreparedDB(rp, nbits)):
        idx, d, cum = H.hamming_topk(qp, db, nbits, k, want_cum=True)
        dm = ranking.hamming_matrix_u8(q, r)
        ref = torch.stack([(dm < b).sum(1) for b in range(nbits + 2)], dim=1)
        assert torch.equal(cum.cpu().long(), ref)
        ri, rd = ranking.hamming_topk_stable(q, r, k)
        assert torch.equal(idx.cpu().long(), ri) and torch.equal(d.cpu().long(), rd)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_float_knn_select_path_with_ties_across_the_threshold(metric):
    """k <= N/2 takes the radix-select path: duplicated rows put equal scores on both sides of the k-th position;
    the k best must come out in ascending (score, index) order like the stable oracle."""
    g = torch.Generator().manual_seed(9)
    Q, N, D, k = 11, 3000, 8, 700
    q = torch.randint(-2, 3, (Q, D), generator=g).float()
    r = torch.randint(-2, 3, (N, D), generator=g).float()
    r[N // 2:] = r[: N - N // 2].clone()
    gi, gd = get_knn(r, q, k, False, with_faiss=False, distance_metric=metric)
    sd, si = ranking.knn_stable(r, q, k, metric)
    assert torch.equal(gi.cpu().long(), si.long())
    assert torch.allclose(gd.cpu(), sd, rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------------- reference-executed fixtures
def test_hip_path_against_reference_executed_outputs(gold):
    """ranking_golden.npz `ref_*` keys = what the reference's own functions returned (tests/golden/make_golden.py cuts
    accuracy_calculator.py:31-37,183-231 and get_knn.py:9-24,60-71 out of the reference files and runs them)."""
    for n in cases(gold):
        q, r = t32(gold[n + "/q"]).cuda(), t32(gold[n + "/r"]).cuda()
        ql, rl = t32(gold[n + "/ql"]).cuda(), t32(gold[n + "/rl"]).cuda()
        k, nbits, Q = int(gold[n + "/k"][0]), q.shape[1], q.shape[0]
        calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
        # distances: the reference's fp32 values, bit for bit
        np.testing.assert_array_equal(calc.calc_hamming_dist(q, r).cpu().numpy(), gold[n + "/ref_dist"])
        np.testing.assert_array_equal(calc.label_comparison_fn(ql, rl).cpu().numpy(), gold[n + "/ref_gnd"])
        # ranked lists vs the order the reference's torch.argsort produced: same distance row, same bucket sets
        idx, dk = H.hamming_topk(H.pack_codes(q), H.pack_codes(r), nbits, k)
        order = torch.from_numpy(gold[n + "/ref_argsort"]).long()
        di = torch.from_numpy(gold[n + "/ref_dist"]).round().long()
        for i in range(Q):
            un = order[i][:k]
            assert torch.equal(di[i][un], dk[i].cpu().long())
            assert ranking.bucket_sets(un, di[i][un]) == ranking.bucket_sets(idx[i].cpu().long(), dk[i].cpu().long())
        # the reported metric.  The tight chain is: HIP == oracle's canonical (stable) value to 1e-6 (here), and the
        # oracle with the reference's recorded order == the reference's value (asserted at generation time and in
        # tests/test_oracle_ranking.py).  Between the canonical order and the reference's unstable argsort lies tie noise
        # only, bounded PER CASE: both values must fall inside the interval of mAPs that orderings differing only inside
        # distance buckets can produce (oracle/ranking.map_tie_bounds; degenerate -- a point -- on the tie-free case).
        m = calc.calculate_maphashing(q, ql, r, rl, k)
        m_all = calc.calculate_maphashing(q, ql, r, rl, None)
        assert abs(m - float(gold[n + "/map_stable"][0])) < 1e-6
        for mine, kk, key in ((m, k, "ref_map"), (m_all, None, "ref_map_all")):
            lo, hi = ranking.map_tie_bounds(gold[n + "/ref_dist"].round(), gold[n + "/ref_gnd"], kk)
            ref = float(gold[f"{n}/{key}"][0])
            assert lo - 1e-6 <= mine <= hi + 1e-6 and lo - 1e-6 <= ref <= hi + 1e-6, (n, kk, lo, mine, ref, hi)
        assert abs(calc.calculate_bit_balance(r) - gold[n + "/ref_bit_balance"][0]) < 1e-6
        assert abs(calc.calculate_worst_bit_balance(r) - gold[n + "/ref_bit_balance"][1]) < 1e-6
        # get_knn, both source modes: the reference's inner products exactly, its index sets per complete bucket
        kk = gold[n + "/ref_knn_idx"].shape[1]
        for same, qq, ki, kd in ((False, q, "ref_knn_idx", "ref_knn_ip"), (True, r[:Q], "ref_selfknn_idx", "ref_selfknn_ip")):
            i_h, d_h = get_knn(r, qq, kk, same, with_faiss=False, distance_metric="hamming")
            assert i_h.dtype == torch.int64 and d_h.dtype == torch.float32
            np.testing.assert_array_equal(d_h.cpu().numpy(), gold[f"{n}/{kd}"])
            ref_i, ref_d = torch.from_numpy(gold[f"{n}/{ki}"]).long(), torch.from_numpy(gold[f"{n}/{kd}"])
            for i in range(Q):
                assert ranking.bucket_sets(i_h[i].cpu(), d_h[i].cpu()) == ranking.bucket_sets(ref_i[i], ref_d[i])


def test_tie_free_case_equals_the_reference_exactly(gold):
    n = "tiefree_q8_n60_b128"
    q, r = t32(gold[n + "/q"]).cuda(), t32(gold[n + "/r"]).cuda()
    ql, rl = t32(gold[n + "/ql"]).cuda(), t32(gold[n + "/rl"]).cuda()
    k = int(gold[n + "/k"][0])
    idx, _ = H.hamming_topk(H.pack_codes(q), H.pack_codes(r), 128, k)
    np.testing.assert_array_equal(idx.cpu().numpy(), gold[n + "/ref_argsort"][:, :k])      # the reference's own order
    calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
    assert abs(calc.calculate_maphashing(q, ql, r, rl, k) - gold[n + "/ref_map"][0]) < AP_TOL
    assert abs(calc.calculate_maphashing(q, ql, r, rl, None) - gold[n + "/ref_map_all"][0]) < AP_TOL
    i_h, d_h = get_knn(r, q, 20, False, with_faiss=False, distance_metric="hamming")
    np.testing.assert_array_equal(i_h.cpu().numpy(), gold[n + "/ref_knn_idx"][:, :20])


def test_label_comparison_other_branches_match_reference(gold):
    calc = CustomCalculator(k=5, distance_metric="hamming", with_faiss=False)
    ql, rl = torch.from_numpy(gold["classid/ql"]).cuda(), torch.from_numpy(gold["classid/rl"]).cuda()
    np.testing.assert_array_equal(calc.label_comparison_fn(ql, rl).cpu().numpy(), gold["classid/ref_gnd"])
    ql3, knn = t32(gold["mixed/ql"]).cuda(), t32(gold["mixed/knn_labels"]).cuda()
    np.testing.assert_array_equal(calc.label_comparison_fn(ql3[:, None], knn).cpu().numpy(), gold["mixed/ref_gnd"])
    # float k-NN against the reference's get_knn_torch (cdist / matmul + topk): same neighbours (no ties in these)
    for metric in ("l2", "cosine"):
        i_h, d_h = get_knn(torch.from_numpy(gold[f"float_{metric}/r"]), torch.from_numpy(gold[f"float_{metric}/q"]), 20,
                           False, with_faiss=False, distance_metric=metric)
        np.testing.assert_array_equal(i_h.cpu().numpy(), gold[f"float_{metric}/idx"])
        np.testing.assert_allclose(d_h.cpu().numpy(), gold[f"float_{metric}/dist"], rtol=2e-5, atol=2e-6)


def test_256_bit_codes_cannot_wrap_a_byte_distance():
    """Complementary 256-bit codes are 256 apart: a uint8 distance would read 0 (the nearest value).  Refused."""
    q = torch.ones(2, 256)
    r = -torch.ones(3, 256)
    calc = CustomCalculator(k=2, distance_metric="hamming", with_faiss=False)
    with pytest.raises(ValueError, match="nbits <= 255"):
        calc.calc_hamming_dist(q, r)
    with pytest.raises(ValueError, match="nbits <= 255"):
        H.hamming_dist(H.pack_codes(q.cuda()), H.pack_codes(r.cuda()))
    d = calc.calc_hamming_dist(q[:, :255], r[:, :255])                   # 255 bits: the largest distance fits
    assert (d == 255).all()
    d = H.hamming_dist(H.pack_codes(q[:, :200].cuda()), H.pack_codes(r[:, :200].cuda()), nbits=200)
    assert (d == 200).all()


def test_l2_values_follow_the_selected_reference_backend(gold):
    """get_knn.py: faiss IndexFlatL2 (with_faiss=True, the default) returns SQUARED L2, torch.cdist true L2."""
    q, r = torch.from_numpy(gold["float_l2/q"]), torch.from_numpy(gold["float_l2/r"])
    i_t, d_t = get_knn(r, q, 20, False, with_faiss=False, distance_metric="l2")
    i_f, d_f = get_knn(r, q, 20, False, with_faiss=True, distance_metric="l2")
    assert torch.equal(i_t, i_f)
    np.testing.assert_allclose(d_f.cpu().numpy(), d_t.cpu().numpy() ** 2, rtol=1e-6)
    np.testing.assert_allclose(d_t.cpu().numpy(), gold["float_l2/dist"], rtol=2e-5, atol=2e-6)


# ---------------------------------------------------------------------------------- windowed kernel (rank2.hip)
def _spread_codes(Q, N, nbits, seed):
    """Distances of every query spread over (nearly) all bins: row j differs from a base code in j % (nbits + 1) bits,
    queries differ from it in a few more -- the ranking needs several 32-bin windows."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(0, 2, (nbits,), generator=g).float() * 2 - 1
    r = base.repeat(N, 1)
    flips = torch.randint(0, nbits + 1, (N,), generator=g)
    for j in range(N):
        pos = torch.randperm(nbits, generator=g)[:int(flips[j])]
        r[j, pos] *= -1
    q = base.repeat(Q, 1)
    for i in range(Q):
        pos = torch.randperm(nbits, generator=g)[:i % 7]
        q[i, pos] *= -1
    return q, r


@pytest.mark.parametrize("variant", ["256", "64", "0"])
@pytest.mark.parametrize("Q,N,nbits,k", [(9, 3000, 64, 2500), (5, 3000, 128, 3000), (33, 1000, 64, 37), (6, 257, 32, 257),
                                          (4100, 700, 64, 200)])
def test_window_kernel_variants_and_multi_window_rankings(diag, variant, Q, N, nbits, k):
    """WV_TOPK_V2 pins the implementation: 256 / 64 threads per query of the windowed kernel, 0 = first-generation
    kernel.  Spread distances force the window to slide; lists, distance rows and histograms must not change."""
    diag.setenv("WV_TOPK_V2", variant)
    q, r = (_spread_codes(Q, N, nbits, seed=N + nbits) if Q < 100 else synth.random_codes(Q, N, nbits, seed=5))
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    ref_idx, ref_d = ranking.hamming_topk_stable(q[:64], r, k)
    for db in (rp, H.PreparedDB(rp, nbits)):
        idx, d = H.hamming_topk(qp, db, nbits, k, idx_offset=1000)
        assert torch.equal(idx[:64].cpu().long() - 1000, ref_idx) and torch.equal(d[:64].cpu().long(), ref_d)
        idx2, d2, cum = H.hamming_topk(qp, db, nbits, k, want_cum=True)
        assert torch.equal(idx2 + 1000, idx) and torch.equal(d2, d)
        dm = ranking.hamming_matrix_u8(q[:64], r)
        ref_cum = torch.stack([(dm < b).sum(1) for b in range(nbits + 2)], dim=1)
        assert torch.equal(cum[:64].cpu().long(), ref_cum)
        if Q > 64:                                               # every query: sorted, consistent with the matrix
            full = H.hamming_dist(qp, rp, nbits=nbits)
            assert torch.equal(torch.gather(full, 1, (idx - 1000).long()), d)
            assert (d[:, 1:] >= d[:, :-1]).all()
            same = d[:, 1:] == d[:, :-1]
            assert (idx[:, 1:][same] > idx[:, :-1][same]).all()


@pytest.mark.parametrize("Q,N,nbits,k,G", [(37, 11000, 64, 3000, 8), (19, 999, 16, 999, 3), (4100, 5000, 64, 1200, 8),
                                           (21, 20000, 128, 5000, 7), (9, 40, 32, 7, 5)])
@pytest.mark.parametrize("prepared", [False, True])
def test_two_step_sharded_search_on_one_gpu(Q, N, nbits, k, G, prepared):
    """The steps every rank of wvhash/parallel.py runs, for G shards on one GPU: wv_hamming_hist per shard -> sum of the
    histograms (the all-reduce) -> global k-th distance -> prefix length -> wv_hamming_topk_rows16 per shard ->
    wv_topk_merge_cum.  Equal to the unsharded ranking; histograms and 16-bit lists equal the oracle's."""
    q, r = (synth.random_codes(Q, N, nbits, seed=N + G) if Q > 100 else _spread_codes(Q, N, nbits, seed=N + G))
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    full_idx, full_d = H.hamming_topk(qp, rp, nbits, k)
    per = (N + G - 1) // G
    shards = []
    for g in range(G):
        lo, hi = min(N, g * per), min(N, (g + 1) * per)
        rows = rp[lo:hi].contiguous()
        shards.append((lo, hi, (H.PreparedDB(rows, nbits) if prepared else rows) if hi > lo else None))
    cums = []
    for lo, hi, db in shards:
        cum = H.hamming_hist(qp, db, nbits) if db is not None else torch.zeros((Q, nbits + 2), dtype=torch.int32, device="cuda")
        if db is not None:
            dm = ranking.hamming_matrix_u8(q[:16], r[lo:hi])
            ref = torch.stack([(dm < b).sum(1) for b in range(nbits + 2)], dim=1)
            assert torch.equal(cum[:16].cpu().long(), ref)
        cums.append(cum)
    cum = torch.stack(cums)
    T = (cum.sum(0)[:, 1:] >= k).int().argmax(dim=1)
    need = int(torch.gather(cum, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max().item())
    send = max(1, min(min(k, per), need))
    lists = []
    for lo, hi, db in shards:
        loc = torch.zeros((Q, send), dtype=torch.int16, device="cuda")
        w = min(send, hi - lo)
        if w > 0:
            loc[:, :w] = H.hamming_topk_rows16(qp, db, nbits, w)
            ref_i, _ = ranking.hamming_topk_stable(q[:16], r[lo:hi], w)
            assert torch.equal((loc[:16, :w].cpu().long() & 0xffff), ref_i)
        lists.append(loc)
    mi, md = H.topk_merge_cum(torch.stack(lists), cum, per, k, nbits)
    assert torch.equal(mi, full_idx) and torch.equal(md, full_d)
    assert send < min(k, per) or G == 1 or k >= N            # the trimmed prefix really is shorter than the full list
    # the hinted one-pass form: list prefix and complete histograms from ONE kernel per shard, the check of the prefix
    # length done by the merge (need_out = the longest prefix any shard owed) -- no histogram sum beforehand
    for hint, exact in ((send, True), (max(1, send - 1), send == 1)):
        hint = min(hint, min(k, per))
        l2, c2 = [], []
        for lo, hi, db in shards:
            loc = torch.zeros((Q, hint), dtype=torch.int16, device="cuda")
            c = torch.zeros((Q, nbits + 2), dtype=torch.int32, device="cuda")
            w = min(hint, hi - lo)
            if w > 0:
                rows, c = H.hamming_shard_prefix(qp, db, nbits, w)
                loc[:, :w] = rows
                assert torch.equal(rows, H.hamming_topk_rows16(qp, db, nbits, w))
            l2.append(loc)
            c2.append(c)
        assert torch.equal(torch.stack(c2), cum)
        owed = torch.zeros(1, dtype=torch.int32, device="cuda")
        mi2, md2 = H.topk_merge_cum(torch.stack(l2), cum, per, k, nbits, need_out=owed)
        assert int(owed.item()) == need
        assert (int(owed.item()) <= hint) == exact
        if exact:
            assert torch.equal(mi2, full_idx) and torch.equal(md2, full_d)


@pytest.mark.parametrize("Q,N,k,Lc", [(64, 25000, 5000, 38), (37, 3000, 3000, 20), (9, 130, 7, 64)])
def test_ap_of_list_prefixes_equals_ap_of_shorter_lists(Q, N, k, Lc):
    """wv_map_at_k_ld over the first k' entries of longer lists (row pitch > k') = AP of lists ranked at k' (what lets
    evaluate_multi_k rank once); and the oracle's within 1e-6."""
    ql, rl = synth.multi_hot_labels(Q, Lc, 0.10, 1), synth.multi_hot_labels(N, Lc, 0.10, 2)
    q, r = synth.structured_codes(ql, 64, 3, 4), synth.structured_codes(rl, 64, 3, 5)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    idx, _ = H.hamming_topk(qp, rp, 64, k)
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    for kk in (k, max(1, k // 2), max(1, k // 7)):
        short, _ = H.hamming_topk(qp, rp, 64, kk)
        ap_prefix, n_prefix = H.map_at_k(idx, qlp, rlp, k=kk)
        ap_short, n_short = H.map_at_k(short, qlp, rlp)
        assert torch.equal(ap_prefix, ap_short) and torch.equal(n_prefix, n_short)
    _, ap_ref = ranking.calculate_maphashing(q[:12], ql[:12], r, rl, k, stable=True, return_per_query=True)
    np.testing.assert_allclose(H.map_at_k(idx, qlp, rlp)[0][:12].cpu().numpy(), ap_ref, atol=AP_TOL)


@pytest.mark.parametrize("variant", [None, "256", "64"])
@pytest.mark.parametrize("Q,N,nbits,k,Lc,spread", [(2048, 25000, 64, 5000, 38, False), (37, 3000, 64, 2500, 24, True),
                                                   (5, 3000, 128, 3000, 64, True), (33, 1000, 64, 37, 5, True),
                                                   (6, 257, 32, 257, 1, True), (4100, 700, 64, 200, 38, False),
                                                   (11, 32768, 64, 8192, 38, False), (3, 4096, 16, 2048, 10, False),
                                                   (41, 14653, 128, 5000, 80, False), (9, 2000, 64, 700, 128, False),
                                                   # mAP@ALL, the reference's published setting (k = database size: MIRFLICKR 19,581,
                                                   # RESULTS.md:412-418): the AP walk runs in chunks of 32 rounds, any k
                                                   (9, 19581, 64, 19581, 24, False), (5, 25000, 64, 25000, 38, False),
                                                   (3, 32768, 64, 32639, 38, False), (3, 40000, 128, 40000, 80, False),
                                                   (2, 117218, 128, 117218, 80, False)])
def test_fused_map_at_k_equals_ranking_then_ap(diag, variant, Q, N, nbits, k, Lc, spread):
    """wv_hamming_map_at_k (list built and evaluated in LDS, never written) against wv_hamming_topk + wv_map_at_k: AP and
    hit counts of every query.  256 threads per query use the AP kernel's summation order -- bit-identical; one wave per
    query sums in another order (fp64 partial sums: equal after the final fp32 rounding up to one ulp)."""
    if variant is not None:
        diag.setenv("WV_TOPK_V2", variant)
    ql, rl = synth.multi_hot_labels(Q, Lc, 0.12, 11), synth.multi_hot_labels(N, Lc, 0.12, 12)
    if spread:
        q, r = _spread_codes(Q, N, nbits, seed=N + nbits)
    else:
        q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    labels = H.PreparedLabels(rlp)
    assert labels.ok
    fused = H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)
    idx, _ = H.hamming_topk(qp, prep, nbits, k, want_dist=False)
    ap_ref, nrel_ref = H.map_at_k(idx, qlp, rlp)
    if fused is None:                                            # outside the fused kernel: must be one of the stated limits
        assert variant == "64" and N > 4096
        return
    ap, nrel = fused
    assert torch.equal(nrel, nrel_ref)
    if variant == "64" or (variant is None and N <= 4096 and Q >= 4096):
        assert (ap - ap_ref).abs().max().item() <= 1.2e-7
    else:
        assert torch.equal(ap, ap_ref)
    # and against the oracle's AP on a few queries
    ref_idx, _ = ranking.hamming_topk_stable(q[:8], r, k)
    for i in range(min(Q, 8)):
        rel = ((rl[ref_idx[i]] * ql[i]).sum(1) > 0).double()
        hits = rel.cumsum(0)
        want = float((rel * hits / torch.arange(1, k + 1).double()).sum() / rel.sum()) if rel.sum() > 0 else 0.0
        assert abs(float(ap[i]) - want) < 1e-6


def test_fused_map_at_k_refuses_what_it_cannot_do():
    q, r = synth.random_codes(4, 300, 64, seed=1)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), 64)
    wide = H.PreparedLabels(H.pack_labels(synth.multi_hot_labels(300, 130, 0.1, 1).cuda()))      # three label words per row
    assert not wide.ok
    assert H.hamming_map_at_k(qp, prep, wide, H.pack_labels(synth.multi_hot_labels(4, 130, 0.1, 2).cuda()), 64, 10) is None
    big = H.PreparedLabels(torch.zeros((40000, 1), dtype=torch.int64, device="cuda"))               # beyond one kernel launch:
    assert big.ok and len(big.parts) == 2                                                           # two virtual shards
    huge = H.PreparedLabels(torch.zeros((64 * 32768 + 1, 1), dtype=torch.int64, device="cuda"))     # beyond 64 virtual shards
    assert not huge.ok


@pytest.mark.parametrize("Q,N,nbits,k,G,Lc", [(37, 11000, 64, 3000, 8, 38), (19, 999, 16, 999, 3, 10), (4100, 5000, 64, 1200, 8, 38),
                                              (21, 20000, 128, 5000, 7, 64), (9, 40, 32, 7, 5, 3), (64, 25000, 64, 5000, 8, 38),
                                              (33, 117218, 128, 5000, 8, 80),
                                              # mAP@ALL (k = N): every shard sends the relevance string of its whole ranking
                                              (5, 117218, 128, 117218, 8, 80), (7, 19581, 64, 19581, 2, 24), (4, 3000, 64, 3000, 8, 38)])
def test_sharded_map_from_relevance_strings_equals_unsharded(Q, N, nbits, k, G, Lc):
    """What the ranks of sharded_hamming_map_at_k run, for G shards on one GPU: wv_hamming_shard_relbits per shard (relevance
    string of the prefix + histograms) -> wv_merge_relbits_map.  AP and hit counts must equal map_at_k of the unsharded
    ranking bit for bit, the reported need must be the true one, and a prefix one entry too short must be flagged."""
    ql, rl = synth.multi_hot_labels(Q, Lc, 0.12, 21), synth.multi_hot_labels(N, Lc, 0.12, 22)
    q, r = (synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)) if Lc >= 10 else \
        _spread_codes(Q, N, nbits, seed=N + G)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    idx, _ = H.hamming_topk(qp, rp, nbits, k, want_dist=False)
    ap_ref, nrel_ref = H.map_at_k(idx, qlp, rlp)
    per = (N + G - 1) // G
    shards = []
    for g in range(G):
        lo, hi = min(N, g * per), min(N, (g + 1) * per)
        shards.append((lo, hi, H.PreparedDB(rp[lo:hi].contiguous(), nbits) if hi > lo else None,
                       H.PreparedLabels(rlp[lo:hi].contiguous()) if hi > lo else None))
    cums = torch.stack([H.hamming_hist(qp, db, nbits) if db is not None else
                        torch.zeros((Q, nbits + 2), dtype=torch.int32, device="cuda") for _, _, db, _ in shards])
    T = (cums.sum(0)[:, 1:] >= k).int().argmax(dim=1)
    need = int(torch.gather(cums, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max().item())
    for send, exact in ((min(min(k, per), need), True), (max(1, need - 1), need == 1)):
        wires = []
        for lo, hi, db, lab in shards:
            wire = torch.zeros((Q, H.relbits_wire_words(send, nbits)), dtype=torch.int64, device="cuda")
            if db is not None:
                assert H.hamming_shard_relbits(qp, db, lab, qlp, nbits, min(send, hi - lo), wire=wire, kin=send) is wire
            wires.append(wire)
        wires = torch.stack(wires)                                   # [G, Q, histogram | relevance string]
        assert torch.equal(H.wire_histograms(wires, nbits), cums)
        owed = torch.zeros(1, dtype=torch.int32, device="cuda")
        ap, nrel = H.merge_relbits_map(wires, send, k, nbits, need_out=owed)
        assert int(owed.item()) == need and (need <= send) == exact
        if exact:
            assert torch.equal(nrel, nrel_ref) and torch.equal(ap, ap_ref)


@pytest.mark.parametrize("Q,N,nbits,k,Lc", [(33, 117218, 128, 5000, 80), (7, 40000, 64, 3000, 38), (5, 70001, 32, 8000, 5)])
def test_databases_beyond_32768_rows_take_virtual_shards(Q, N, nbits, k, Lc):
    """A PreparedDB of more than 32,768 rows is ranked as contiguous virtual shards through the windowed kernel (histograms
    -> prefix length -> 16-bit lists -> merge): lists and distances identical to the first-generation kernel on the whole
    database; wv_hamming_map_at_k's counterpart (relevance strings of the virtual shards) identical to ranking + AP."""
    ql, rl = synth.multi_hot_labels(Q, Lc, 0.1, 31), synth.multi_hot_labels(N, Lc, 0.1, 32)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    prep = H.PreparedDB(rp, nbits)
    assert prep.parts and len(prep.parts) == -(-N // 32768) and sum(p_.N for p_ in prep.parts) == N
    idx0, d0 = H.hamming_topk(qp, rp, nbits, k)                   # raw tensor: first-generation kernel
    idx1, d1 = H.hamming_topk(qp, prep, nbits, k, idx_offset=7)
    assert torch.equal(idx1 - 7, idx0) and torch.equal(d1, d0)
    ref_idx, ref_d = ranking.hamming_topk_stable(q[:4], r, k)
    assert torch.equal(idx1[:4].cpu().long() - 7, ref_idx) and torch.equal(d1[:4].cpu().long(), ref_d)
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    ap_ref, nrel_ref = H.map_at_k(idx0, qlp, rlp)
    got = H.hamming_map_at_k(qp, prep, H.PreparedLabels(rlp), qlp, nbits, k)
    assert got is not None and torch.equal(got[1], nrel_ref) and torch.equal(got[0], ap_ref)
    calc = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False)
    m = calc.calculate_maphashing(q.cuda(), ql.cuda(), r.cuda(), rl.cuda(), k)
    assert abs(m - float(ap_ref.double().mean())) < 1e-9


@pytest.mark.parametrize("Q,N,nbits,k,lc", [(64, 25000, 64, 5000, 38), (33, 5717, 16, 5717, 20), (17, 3000, 128, 700, 80)])
def test_host_rank_twins_equal_the_kernels(Q, N, nbits, k, lc):
    """csrc/host_rank.cpp (CustomCalculator(device='cpu')) and the gfx950 kernels return the same bits: packed words, per-bit
    counts, distance matrix, ranked lists + distance rows, running hit counts -- and the average precision, whose fp32
    quotients the twin sums in k_map_at_k's order (thread, wave butterfly, waves)."""
    from wvhash.engine import hamming_host as HH
    ql, rl = synth.multi_hot_labels(Q, lc, 0.1, 21), synth.multi_hot_labels(N, lc, 0.1, 22)
    q, r = synth.structured_codes(ql, nbits, 3, 23), synth.structured_codes(rl, nbits, 3, 24)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    qh, rh = HH.pack_codes(q), HH.pack_codes(r)
    assert torch.equal(qp.cpu(), qh) and torch.equal(rp.cpu(), rh)
    assert torch.equal(H.bit_counts(rp, nbits).cpu(), HH.bit_counts(rh, nbits))
    assert torch.equal(H.hamming_dist(qp, rp, nbits).cpu(), HH.hamming_dist(qh, rh, nbits))
    idx, d = H.hamming_topk(qp, rp, nbits, k)
    idx_h, d_h = HH.hamming_topk(qh, rh, nbits, k)
    assert torch.equal(idx.cpu(), idx_h) and torch.equal(d.cpu(), d_h)
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    qlh, rlh = HH.pack_labels(ql), HH.pack_labels(rl)
    assert torch.equal(qlp.cpu(), qlh) and torch.equal(rlp.cpu(), rlh)
    ap, nrel = H.map_at_k(idx, qlp, rlp)
    ap_h, nrel_h = HH.map_at_k(idx_h, qlh, rlh)
    assert torch.equal(nrel.cpu(), nrel_h) and torch.equal(ap.cpu(), ap_h)          # bit-identical floats
    assert torch.equal(H.hit_prefix(idx, qlp, rlp).cpu(), HH.hit_prefix(idx_h, qlh, rlh))
    # the two calculators report the same metric
    from wvhash.engine import CustomCalculator
    m_gpu = CustomCalculator(k=k, distance_metric="hamming", with_faiss=False).calculate_maphashing(q, ql, r, rl, k)
    m_cpu = CustomCalculator(k=k, device="cpu", distance_metric="hamming", with_faiss=False).calculate_maphashing(q, ql, r, rl, k)
    assert abs(m_gpu - m_cpu) < 1e-9


@pytest.mark.parametrize("k", [32639, 32640, 32768])
def test_list_length_at_the_windowed_kernels_limit(k):
    """The windowed kernel counts list positions in BYTES in 16-bit cells: it takes k <= 32,639 (2 (k + 128) < 65,536); one more
    entry and the launch goes to the first-generation kernel.  Same lists either side of the limit (N = 32,768 rows: the
    largest shard), prepared and plain database."""
    Q, N, nbits = 3, 32768, 64
    q, r = synth.random_codes(Q, N, nbits, seed=91)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    ref_idx, ref_d = ranking.hamming_topk_stable(q, r, k)
    for db in (rp, H.PreparedDB(rp, nbits)):
        idx, d = H.hamming_topk(qp, db, nbits, k)
        assert torch.equal(idx.cpu().long(), ref_idx) and torch.equal(d.cpu().long(), ref_d)

"""Pins oracle/ranking.py (CPU only) to outputs of the REFERENCE'S OWN ranking code: tests/golden/ranking_golden.npz
holds what accuracy_calculator.py:31-37,183-231 and get_knn.py:9-24,60-71 returned when tests/golden/make_golden.py
cut those functions out of the reference files and ran them (keys `<case>/ref_*`), plus the behavioural known answers
of /root/reference/studies/measure_random_baseline.py (constant codes -> mAP = relevance-driven floor)."""
import numpy as np
import pytest
import torch

from oracle import ranking


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/ranking_golden.npz")


def case_names(g):
    return sorted({k.split("/")[0] for k in g.files if k.endswith("/topk_idx")})


def load_case(g, n):
    t = lambda k, dt=torch.float32: torch.from_numpy(g[f"{n}/{k}"].astype(np.float32)).to(dt)
    return t("q"), t("r"), t("ql"), t("rl"), int(g[f"{n}/k"][0])


def test_golden_distances_topk_and_map(gold):
    names = case_names(gold)
    assert len(names) == 5
    for n in names:
        q, r, ql, rl, k = load_case(gold, n)
        d = ranking.calc_hamming_dist(q, r)
        np.testing.assert_array_equal(d.numpy(), gold[f"{n}/dist"])
        idx, dk = ranking.hamming_topk_stable(q, r, k)
        np.testing.assert_array_equal(idx.numpy(), gold[f"{n}/topk_idx"])
        np.testing.assert_array_equal(dk.numpy(), gold[f"{n}/topk_dist"])
        m, ap = ranking.calculate_maphashing(q, ql, r, rl, k, stable=True, return_per_query=True)
        np.testing.assert_allclose(ap, gold[f"{n}/ap_stable"], rtol=0, atol=0)
        assert m == gold[f"{n}/map_stable"][0]
        bb = [ranking.calculate_bit_balance(r), ranking.calculate_worst_bit_balance(r)]
        np.testing.assert_array_equal(bb, gold[f"{n}/bit_balance"])


def test_hamming_formula_is_exact_integer_popcount(gold):
    for n in case_names(gold):
        q, r, *_ = load_case(gold, n)
        d = ranking.calc_hamming_dist(q, r)
        brute = ((q[:, None, :] != r[None, :, :]).sum(-1)).float()
        assert torch.equal(d, brute)


def test_unstable_reference_order_differs_only_inside_ties(gold):
    """argsort(stable=False) (the reference's literal call) vs the canonical order: same sorted
    distances, same index set in every complete bucket."""
    for n in case_names(gold):
        q, r, ql, rl, k = load_case(gold, n)
        d = ranking.hamming_matrix_u8(q, r)
        idx_s, dk_s = ranking.hamming_topk_stable(q, r, k)
        for i in range(q.shape[0]):
            un = torch.argsort(d[i].float())[:k]
            assert torch.equal(d[i][un], dk_s[i])
            assert ranking.bucket_sets(un, d[i][un]) == ranking.bucket_sets(idx_s[i], dk_s[i])
        # the reference-order mAP stored in the fixture stays within tie noise of the canonical one
        assert abs(gold[f"{n}/map_ref"][0] - gold[f"{n}/map_stable"][0]) < 0.05


def test_constant_codes_give_relevance_floor():
    """measure_random_baseline.py:110-115: every code identical -> ranking = database order."""
    ql = ranking.make_labels(20, 38, 0.1, 1)
    rl = ranking.make_labels(400, 38, 0.1, 2)
    q, r = torch.ones(20, 64), torch.ones(400, 64)
    m = ranking.calculate_maphashing(q, ql, r, rl, 100, stable=True)
    rel = ranking.label_comparison_fn(ql, rl).float()[:, :100]
    expect = 0.0
    for i in range(20):
        hits = torch.where(rel[i] == 1)[0].float() + 1
        if len(hits):
            expect += (torch.arange(1, len(hits) + 1).float() / hits).mean().item()
    assert abs(m - expect / 20) < 1e-12


def test_zero_hit_queries_count_as_zero_and_topk_none_means_all():
    q, r = ranking.make_codes(3, 50, 16, 3)
    ql = torch.zeros(3, 5); ql[:, 0] = 1
    rl = torch.zeros(50, 5); rl[:, 1] = 1
    assert ranking.calculate_maphashing(q, ql, r, rl, 10) == 0.0
    rl[7, 0] = 1
    a = ranking.calculate_maphashing(q, ql, r, rl, None, stable=True)
    b = ranking.calculate_maphashing(q, ql, r, rl, 50, stable=True)
    assert a == b > 0


def test_label_comparison_branches():
    a = torch.tensor([[1., 0, 1], [0, 1, 0]])
    b = torch.tensor([[0., 0, 1], [0, 0, 0], [1, 1, 0]])
    assert ranking.label_comparison_fn(a, b).tolist() == [[True, False, True], [False, False, True]]
    ids_q, ids_r = torch.tensor([3, 5]), torch.tensor([5, 3, 3])
    assert ranking.label_comparison_fn(ids_q, ids_r).tolist() == [[False, True, True], [True, False, False]]
    knn = b[torch.tensor([[0, 2], [1, 2]])]            # [Q, k, Lc] vs [Q, 1, Lc]
    assert ranking.label_comparison_fn(a[:, None], knn).tolist() == [[True, True], [False, True]]


def test_get_knn_shapes_and_same_source(gold):
    q, r, *_ = load_case(gold, "rand_q16_n500_b32")
    idx, dist = ranking.get_knn(r, q, 10, False, distance_metric="hamming")
    assert idx.shape == (16, 10) and dist.shape == (16, 10) and idx.dtype == torch.int64
    np.testing.assert_array_equal(dist.numpy(), gold["rand_q16_n500_b32/knn_ip"][:, :10])
    idx2, dist2 = ranking.get_knn(r, r[:4], 5, True, distance_metric="hamming")
    assert idx2.shape == (4, 5) and (dist2 <= 32).all()
    for m in ("l2", "cosine"):
        d, i = ranking.get_knn_torch(torch.from_numpy(gold[f"float_{m}/r"]), torch.from_numpy(gold[f"float_{m}/q"]), 20, m)
        np.testing.assert_array_equal(i.numpy(), gold[f"float_{m}/idx"])
        np.testing.assert_allclose(d.numpy(), gold[f"float_{m}/dist"], rtol=1e-6)


# ---------------------------------------------------------------------------------- reference-executed fixtures
def _ap_on_order(order, gnd_row, k):
    """AP formula of accuracy_calculator.py:223-229 on a given ranking."""
    t = gnd_row[order][:k]
    n = int(t.sum())
    if n == 0:
        return 0.0
    pos = torch.where(t == 1)[0].float() + 1.0
    return torch.mean(torch.arange(1, n + 1).float() / pos).item()


def test_oracle_reproduces_the_reference_executed_outputs(gold):
    for n in case_names(gold):
        q, r, ql, rl, k = load_case(gold, n)
        ref_d = torch.from_numpy(gold[f"{n}/ref_dist"])
        assert torch.equal(ranking.calc_hamming_dist(q, r), ref_d)                       # :183-186
        gnd = ranking.label_comparison_fn(ql, rl)
        np.testing.assert_array_equal(gnd.numpy(), gold[f"{n}/ref_gnd"])                  # :31-37
        np.testing.assert_array_equal([ranking.calculate_bit_balance(r), ranking.calculate_worst_bit_balance(r)],
                                      gold[f"{n}/ref_bit_balance"])                       # :188-200
        # the reference's own argsort order, replayed through the AP formula, gives the reference's mAP (:203-231)
        order = torch.from_numpy(gold[f"{n}/ref_argsort"]).long()
        for kk, key in ((k, "ref_map"), (None, "ref_map_all")):
            aps = [_ap_on_order(order[i], gnd[i].float(), kk) for i in range(q.shape[0])]
            assert abs(sum(aps) / len(aps) - gold[f"{n}/{key}"][0]) < 1e-12
        # canonical order = the reference's order up to permutations inside a distance bucket
        idx, dk = ranking.hamming_topk_stable(q, r, k)
        di = ref_d.round().long()
        for i in range(q.shape[0]):
            un = order[i][:k]
            assert torch.equal(di[i][un], dk[i])
            assert ranking.bucket_sets(un, di[i][un]) == ranking.bucket_sets(idx[i], dk[i])
        # k-NN: same inner products, same index sets per complete bucket (get_knn.py:9-24, 60-71)
        ref_ip, ref_i = torch.from_numpy(gold[f"{n}/ref_knn_ip"]), torch.from_numpy(gold[f"{n}/ref_knn_idx"]).long()
        d_o, i_o = ranking.knn_stable(r, q, ref_ip.shape[1], "hamming")
        assert torch.equal(d_o, ref_ip)
        for i in range(q.shape[0]):
            assert ranking.bucket_sets(i_o[i], d_o[i]) == ranking.bucket_sets(ref_i[i], ref_ip[i])
        si, sd = ranking.get_knn(r, r[:q.shape[0]], ref_ip.shape[1], True, distance_metric="hamming")
        np.testing.assert_array_equal(sd.numpy(), gold[f"{n}/ref_selfknn_ip"])


def test_tie_free_case_is_exact(gold):
    """Every distance of a query distinct -> the reference's unstable argsort IS the canonical order."""
    n = "tiefree_q8_n60_b128"
    q, r, ql, rl, k = load_case(gold, n)
    d = ranking.hamming_matrix_u8(q, r)
    assert all(len(set(row.tolist())) == r.shape[0] for row in d)
    idx, _ = ranking.hamming_topk_stable(q, r, k)
    np.testing.assert_array_equal(idx.numpy(), gold[f"{n}/ref_argsort"][:, :k])
    assert ranking.calculate_maphashing(q, ql, r, rl, k, stable=True) == gold[f"{n}/ref_map"][0]


def test_label_comparison_other_branches_match_reference(gold):
    ql, rl = torch.from_numpy(gold["classid/ql"]), torch.from_numpy(gold["classid/rl"])
    np.testing.assert_array_equal(ranking.label_comparison_fn(ql, rl).numpy(), gold["classid/ref_gnd"])
    ql3 = torch.from_numpy(gold["mixed/ql"].astype(np.float32))
    knn = torch.from_numpy(gold["mixed/knn_labels"].astype(np.float32))
    np.testing.assert_array_equal(ranking.label_comparison_fn(ql3[:, None], knn).numpy(), gold["mixed/ref_gnd"])


def test_reference_map_lies_inside_the_tie_bounds_of_its_own_distances(gold):
    """The only freedom between the reference's unstable argsort and the canonical order is the order inside distance
    buckets; map_tie_bounds gives the exact interval of mAP@k that freedom spans.  The reference-executed values and the
    canonical ones lie inside it for every fixture, and the interval is a point where no ties exist."""
    for n in ["rand_q5_n64_b16", "rand_q16_n500_b32", "struct_q12_n1000_b64", "struct_q8_n777_b128", "tiefree_q8_n60_b128"]:
        k = int(gold[f"{n}/k"][0])
        for kk, key in ((k, "ref_map"), (None, "ref_map_all")):
            lo, hi = ranking.map_tie_bounds(gold[f"{n}/ref_dist"].round(), gold[f"{n}/ref_gnd"], kk)
            assert lo - 1e-6 <= float(gold[f"{n}/{key}"][0]) <= hi + 1e-6      # (the reference sums fp32 quotients)
        lo, hi = ranking.map_tie_bounds(gold[f"{n}/ref_dist"].round(), gold[f"{n}/ref_gnd"], k)
        assert lo - 1e-6 <= float(gold[f"{n}/map_stable"][0]) <= hi + 1e-6
        if n.startswith("tiefree"):
            assert hi - lo < 1e-12

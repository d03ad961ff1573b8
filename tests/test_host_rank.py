"""Host twins of the ranking side (csrc/host_rank.cpp behind CustomCalculator(device='cpu')) on the CPU box, against the
outputs of the REFERENCE's own functions (tests/golden/ranking_golden.npz keys `ref_*`: the bodies of calc_hamming_dist,
label_comparison_fn, per_bit_balance, calculate_maphashing, get_knn were cut out of
/root/reference/main/engine/accuracy_calculator.py:31-37,183-231 and get_knn.py:9-24,60-71 and executed at generation time)
and against the canonical (stable) lists.  No GPU is touched: this is BASELINE config c0's mode ("CPU ... plumbing")."""
import os

import numpy as np
import pytest
import torch

from oracle import ranking
from wvhash import synth
from wvhash.engine import CustomCalculator, get_accuracy_calculator
from wvhash.engine import hamming_host as HH

CASES = ["rand_q5_n64_b16", "rand_q16_n500_b32", "struct_q12_n1000_b64", "struct_q8_n777_b128", "tiefree_q8_n60_b128"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "ranking_golden.npz"))


def _case(gold, n):
    return (torch.from_numpy(gold[f"{n}/q"]), torch.from_numpy(gold[f"{n}/ql"]), torch.from_numpy(gold[f"{n}/r"]),
            torch.from_numpy(gold[f"{n}/rl"]), int(gold[f"{n}/k"][0]))


@pytest.mark.parametrize("n", CASES)
def test_twins_reproduce_the_reference_executed_outputs(gold, n):
    q, ql, r, rl, k = _case(gold, n)
    nbits = q.shape[1]
    qp, rp = HH.pack_codes(q), HH.pack_codes(r)
    # calc_hamming_dist (:183-186): exact integers
    assert torch.equal(HH.hamming_dist(qp, rp, nbits).float(), torch.from_numpy(gold[f"{n}/ref_dist"]))
    # ranking: the canonical order (ascending distance, then row) = the stored stable lists, bit for bit ...
    idx, d = HH.hamming_topk(qp, rp, nbits, k)
    np.testing.assert_array_equal(idx.numpy(), gold[f"{n}/topk_idx"])
    np.testing.assert_array_equal(d.numpy(), gold[f"{n}/topk_dist"])
    # ... which is the reference's own (unstable) argsort up to the order inside a distance bucket
    ref_order = torch.from_numpy(gold[f"{n}/ref_argsort"]).long()[:, :k]
    ref_d = torch.from_numpy(gold[f"{n}/ref_dist"])
    for i in range(q.shape[0]):
        assert torch.equal(torch.gather(ref_d[i], 0, ref_order[i]).to(torch.uint8), d[i])
        assert ranking.bucket_sets(idx[i].long(), d[i]) == ranking.bucket_sets(ref_order[i], torch.gather(ref_d[i], 0, ref_order[i]))
    # AP per query = the oracle's on the stable lists; mAP = the reference's up to its tie noise (equal where tie-free)
    ap, nrel = HH.map_at_k(idx, HH.pack_labels(ql), HH.pack_labels(rl))
    assert np.abs(ap.numpy() - gold[f"{n}/ap_stable"]).max() < 1e-6
    assert abs(float(ap.double().mean()) - float(gold[f"{n}/map_stable"][0])) < 1e-6
    if n.startswith("tiefree"):
        np.testing.assert_array_equal(idx.numpy(), gold[f"{n}/ref_argsort"][:, :k])
        assert abs(float(ap.double().mean()) - float(gold[f"{n}/ref_map"][0])) < 1e-6
    # relevance (:31-37) through the packed words
    gnd = torch.from_numpy(gold[f"{n}/ref_gnd"])
    hits = HH.hit_prefix(idx, HH.pack_labels(ql), HH.pack_labels(rl))
    want = torch.gather(gnd, 1, idx.long()).int().cumsum(1).int()
    assert torch.equal(hits, want) and torch.equal(nrel, want[:, -1])
    # per_bit_balance (:188-200)
    calc = CustomCalculator(k=k, device="cpu", distance_metric="hamming", with_faiss=False)
    np.testing.assert_allclose([calc.calculate_bit_balance(r), calc.calculate_worst_bit_balance(r)], gold[f"{n}/ref_bit_balance"],
                               rtol=0, atol=1e-7)
    # get_knn (get_knn.py:9-24, 60-71): inner products equal; same index sets per score bucket; same-source drops column 0
    ki, kd = calc._host_knn(r, q, gold[f"{n}/ref_knn_ip"].shape[1], False)
    np.testing.assert_array_equal(kd.numpy(), gold[f"{n}/ref_knn_ip"])
    si, sd = calc._host_knn(r, r[:q.shape[0]], gold[f"{n}/ref_selfknn_ip"].shape[1], True)
    np.testing.assert_array_equal(sd.numpy(), gold[f"{n}/ref_selfknn_ip"])


def test_calculator_on_cpu_runs_the_c0_shape_without_a_gpu():
    """BASELINE config c0: VOC-sized database (5,717 codes), 16-bit hash, k = N, everything on the host.  get_accuracy with the
    exclude list of the reference's evaluate.py (BASE_EXCLUDE_METRICS, evaluate.py:39-45) returns maphashing / map / bit
    balances; maphashing equals the oracle's canonical value, `map` is the AP over the k-NN lists of the non-lone queries."""
    Q, N, nbits, lc = 300, 5717, 16, 20
    ql, rl = synth.multi_hot_labels(Q, lc, 0.07, 1), synth.multi_hot_labels(N, lc, 0.07, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    exclude = ["mean_reciprocal_rank", "mean_average_precision", "mean_average_precision_at_r", "precision_at_1", "recall_at_1",
               "r_precision", "rpr", "pr", "pr_rc", "recall_at_1000", "recall_at_100", "recall_at_10", "recall_at_16", "recall_at_20",
               "recall_at_30", "recall_at_32", "recall_at_4", "recall_at_8", "recall_at_2"]
    calc = get_accuracy_calculator(k=N, device=torch.device("cpu"), distance_metric="hamming", with_faiss=False, exclude=exclude)
    assert calc.host and calc.device.type == "cpu"
    out = calc.get_accuracy(q, ql, r, rl, False)
    assert set(out) == {"maphashing", "map", "bit_balance", "worst_bit_balance"}
    want = ranking.calculate_maphashing(q, ql, r, rl, N, stable=True)
    assert abs(out["maphashing"] - want) < 1e-6
    assert 0.0 < out["map"] <= 1.0 and 0.0 <= out["worst_bit_balance"] <= out["bit_balance"] <= 1.0
    # the same numbers from numpy inputs and an int k < N, as measure_random_baseline.py:102-107 calls it
    calc5 = CustomCalculator(k=500, device=torch.device("cpu"), distance_metric="hamming", with_faiss=False)
    m5 = calc5.calculate_maphashing(q.numpy(), ql.numpy(), r.numpy(), rl.numpy(), 500)
    assert abs(m5 - ranking.calculate_maphashing(q, ql, r, rl, 500, stable=True)) < 1e-6


def test_cpu_calculator_refuses_what_it_does_not_cover():
    from wvhash import _lib
    calc = CustomCalculator(k=5, device="cpu", distance_metric="cosine", with_faiss=False)
    x = torch.randn(6, 18)
    idx, _ = calc._host_knn(x, x[:2], 3, False)                  # real-valued embeddings: the float twin (any dimension)
    assert idx[:, 0].tolist() == [0, 1]
    with pytest.raises(ValueError, match="exactly"):
        calc.calculate_maphashing(torch.zeros(2, 16), torch.ones(2, 3), torch.ones(4, 16), torch.ones(4, 3), 2)

"""Host twins of the real-valued k-NN, the attention-pooling head and the hashing tail ("_cpu twins of
each"; csrc/host_knn.cpp, csrc/host_head.cpp) on a box without a GPU: against the stable oracle, the reference-made golden
vectors of the head (tests/golden/head_golden.npz, same tolerance as the kernels: 5e-5) and stock torch."""
import os

import numpy as np
import pytest
import torch

from oracle import ranking
from wvhash import _lib
from wvhash.engine import CustomCalculator
from wvhash.engine.get_knn import knn_float_host

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("metric,name,D", [(_lib.WV_METRIC_IP, "cosine", 8), (_lib.WV_METRIC_L2, "l2", 8),
                                           (_lib.WV_METRIC_IP, "cosine", 20), (_lib.WV_METRIC_L2, "l2", 36),
                                           (_lib.WV_METRIC_IP, "cosine", 3), (_lib.WV_METRIC_L2, "l2", 70)])
def test_knn_float_cpu_is_the_stable_ranking_on_exact_scores(metric, name, D):
    """Integer-valued embeddings: every score is exact in fp32 whatever the summation order, so the stable order is the
    unique answer (ties by ascending index, duplicated rows across the k-th position)."""
    g = torch.Generator().manual_seed(D)
    Q, N, k = 9, 700, 300
    q = torch.randint(-3, 4, (Q, D), generator=g).float()
    r = torch.randint(-3, 4, (N, D), generator=g).float()
    r[N // 2:] = r[: N - N // 2].clone()
    v, i = knn_float_host(r, q, k, metric)
    sd, si = ranking.knn_stable(r, q, k, name)
    assert torch.equal(i.long(), si.long())
    torch.testing.assert_close(v, sd, rtol=3e-7, atol=0)
    if metric == _lib.WV_METRIC_L2:                       # the faiss flavour: same neighbours, squared distances (exact here)
        v2, i2 = knn_float_host(r, q, k, _lib.WV_METRIC_L2_SQUARED)
        d2 = ((q[:, None, :] - r[None, :, :]) ** 2).sum(-1)
        assert torch.equal(i2, i) and torch.equal(v2, torch.gather(d2, 1, i.long()))


def test_knn_float_cpu_random_embeddings_and_errors():
    g = torch.Generator().manual_seed(2)
    q, r = torch.randn(17, 64, generator=g), torch.randn(1000, 64, generator=g)
    v, i = knn_float_host(r, q, 50, _lib.WV_METRIC_IP)
    full = q @ r.t()
    np.testing.assert_allclose(torch.gather(full, 1, i.long()).numpy(), v.numpy(), atol=3e-5)
    assert (v[:, 1:] <= v[:, :-1]).all()
    want = torch.topk(full, 50, dim=1).values
    np.testing.assert_allclose(v.numpy(), want.numpy(), atol=3e-5)
    v, i = knn_float_host(r, q, 1000, _lib.WV_METRIC_L2)            # k = N
    np.testing.assert_allclose(v.numpy(), torch.sort(torch.cdist(q, r), dim=1).values.numpy(), atol=3e-5)
    assert all(sorted(row.tolist()) == list(range(1000)) for row in i[:3])
    lib = _lib.load()
    out_i, out_v = torch.zeros(2, 1, dtype=torch.int32), torch.zeros(2, 1)
    x = torch.zeros(2, 8)
    assert lib.wv_knn_float_cpu(_lib.ptr(x), _lib.ptr(x), 2, 2, 8, 0, 3, _lib.ptr(out_i), _lib.ptr(out_v)) == -22   # k > N
    assert lib.wv_knn_float_cpu(_lib.ptr(x), _lib.ptr(x), 2, 2, 8, 7, 1, _lib.ptr(out_i), _lib.ptr(out_v)) == -22   # metric


def test_cpu_calculator_runs_float_metrics():
    """CustomCalculator(device='cpu') with real-valued embeddings: get_knn's cosine / l2 branches (get_knn.py:63-69) and the
    faiss flavour's squared distances through the host twin."""
    g = torch.Generator().manual_seed(5)
    r, q = torch.randn(300, 32, generator=g), torch.randn(12, 32, generator=g)
    for metric, faiss in (("cosine", False), ("l2", False), ("l2", True)):
        calc = CustomCalculator(k=10, device="cpu", distance_metric=metric, with_faiss=faiss)
        idx, dist = calc._host_knn(r, q, 10, False)
        sd, si = ranking.knn_stable(r, q, 10, metric)
        assert torch.equal(idx, si.long())
        np.testing.assert_allclose(dist.numpy(), (sd * sd if faiss else sd).numpy(), rtol=2e-5, atol=2e-5)
    idx, dist = CustomCalculator(k=5, device="cpu", distance_metric="l2", with_faiss=False)._host_knn(r, r[:7], 5, True)
    assert idx.shape == (7, 5) and not (idx == torch.arange(7)[:, None]).any()       # same source: column 0 (self) stripped


TYPES = {"adv": "cross_attention_advanced", "base": "cross_attention_bottleneck",
         "pooled": "cross_attention_pooled", "decoupled": "cross_attention_decoupled"}


def _build(n, gold):
    """The reference-made golden cases as tests/test_gpu_head.py builds them (seeded state_dict and features; the golden file
    holds the reference module's outputs)."""
    from wvhash import synth
    from wvhash.models import get_fusion_head
    E, heads, nq, B, seed, mean, dec = gold[n + "/meta"].tolist()
    cfg = {"type": TYPES[n.split("_")[0]], "output_dim": E, "num_heads": heads, "num_queries": nq,
           "sub_band_dropout_p": 0.0, "ortho_weight": 0.1}
    if mean:
        cfg["query_pool"] = "mean"
    qs = float(gold[n + "/qscale"][0])
    if dec:
        cfg["query_scale_init"] = qs
    head = get_fusion_head(cfg, [E] * 4)
    head.load_state_dict(synth.head_state(E, nq, "mean" if mean else "concat", seed, query_scale=qs if dec else None))
    return head.eval(), synth.band_features(B, E, seed + 1000)


def test_head_twin_matches_the_reference_module_outputs():
    """All seven golden cases (four head variants, Nq 1 / 4 / 8, E 64 / 384, mean read-out, scaled normalised queries) through
    wv_band_attn_pool_cpu: the same 5e-5 the kernels are held to.  Without the explicit switch a host tensor is refused."""
    gold = np.load(os.path.join(GOLD, "head_golden.npz"))
    names = sorted({k.split("/")[0] for k in gold.files if k.endswith("/meta")})
    assert len(names) == 7
    for n in names:
        head, feats = _build(n, gold)
        with torch.no_grad():
            with pytest.raises(_lib.WvhashUnavailable, match="host_twin"):
                head(feats)                                        # never a silent fallback
            head.host_twin = True
            y = head(feats)
            y2 = head(feats)
        assert torch.equal(y, y2) and y.shape == gold[n + "/out"].shape
        assert np.abs(y.numpy() - gold[n + "/out"]).max() < 5e-5, n
        assert float(head.last_ortho_loss) == 0.0


def test_hash_tail_twin_matches_stock_torch_and_the_golden_codes():
    from wvhash.models import hash_tail
    g = torch.Generator().manual_seed(4)
    B, E, nbits = 37, 384, 64
    fc = torch.nn.Linear(E, nbits, bias=False)
    bn = torch.nn.BatchNorm1d(nbits).eval()
    with torch.no_grad():
        fc.weight.copy_(torch.randn(nbits, E, generator=g) * 0.05)
        bn.weight.copy_(torch.rand(nbits, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(nbits, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(nbits, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(nbits, generator=g) + 0.5)
    fused = torch.randn(B, E, generator=g)
    with pytest.raises(_lib.WvhashUnavailable):
        hash_tail(fused, fc, bn)                                   # host tensors without the explicit switch
    out = hash_tail(fused, fc, bn, want=("logits", "codes", "packed"), host_twin=True)
    with torch.no_grad():
        ref = bn(fc(fused))
    np.testing.assert_allclose(out["logits"].numpy(), ref.numpy(), atol=2e-5)
    sure = ref.abs() > 1e-4
    assert torch.equal(out["codes"][sure], torch.sign(ref)[sure])
    bits = ((out["packed"][:, :, None] >> torch.arange(64)) & 1).reshape(B, -1)[:, :nbits]
    assert torch.equal(bits == 1, out["codes"] > 0)
    # 100-bit codes: two packed words, the second partly filled; Linear with a bias and no BatchNorm
    fc2 = torch.nn.Linear(E, 100)
    out2 = hash_tail(fused, fc2, None, want=("logits", "packed"), host_twin=True)
    with torch.no_grad():
        np.testing.assert_allclose(out2["logits"].numpy(), fc2(fused).numpy(), atol=2e-5)
    assert out2["packed"].shape == (B, 2) and int((out2["packed"][:, 1] >> 36).abs().sum()) == 0

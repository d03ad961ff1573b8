"""Pins oracle/head_torch.py against outputs of the REFERENCE's own head modules
(tests/golden/head_golden.npz, produced by importing
/root/reference/main/models/multi_dino_attention.py by file path in the build container)."""
import numpy as np
import pytest
import torch

from oracle import head_torch
from wvhash import synth


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/head_golden.npz")


def head_cases(g):
    return sorted({k.split("/")[0] for k in g.files if k.endswith("/meta")})


def rebuild(g, n):
    E, heads, nq, B, seed, mean, dec = g[n + "/meta"].tolist()
    qs = float(g[n + "/qscale"][0])
    sd = synth.head_state(E, nq, "mean" if mean else "concat", seed, query_scale=qs if dec else None)
    assert synth.state_sha(sd) == bytes(g[n + "/sha"]).hex(), "seeded weights changed"
    return sd, synth.band_features(B, E, seed + 1000), heads, ("mean" if mean else "concat"), bool(dec)


def test_restatement_matches_reference_outputs(gold):
    names = head_cases(gold)
    assert len(names) == 7
    for n in names:
        sd, feats, heads, pool, dec = rebuild(gold, n)
        y, w = head_torch.band_attn_pool(feats, sd, heads, pool, normalize_queries=dec, return_weights=True)
        np.testing.assert_allclose(y.numpy(), gold[n + "/out"], atol=1e-5, rtol=0)
        np.testing.assert_allclose(w.numpy(), gold[n + "/attn_w"], atol=2e-7, rtol=0)


def test_fp64_restatement_bounds_fp32_error(gold):
    sd, feats, heads, pool, dec = rebuild(gold, "adv_e384_nq4")
    y64 = head_torch.band_attn_pool(feats, sd, heads, pool, dtype=torch.float64)
    assert np.abs(y64.numpy() - gold["adv_e384_nq4/out"]).max() < 1e-5


def test_hash_tail_matches_reference_modules(gold):
    tail = synth.hash_tail_state(384, 64, seed=21)
    assert synth.state_sha(tail) == bytes(gold["tail/sha"]).hex()
    fused = torch.from_numpy(gold["adv_e384_nq4/out"])
    logits = head_torch.hash_tail(fused, tail["hash_fc.weight"], tail["bn.weight"], tail["bn.bias"],
                                  tail["bn.running_mean"], tail["bn.running_var"], return_logits=True)
    np.testing.assert_allclose(logits.numpy(), gold["tail/logits"], atol=2e-6)
    far = np.abs(gold["tail/logits"]) > 1e-4
    assert np.array_equal(torch.sign(logits).numpy()[far], gold["tail/codes"][far])


def test_ortho_loss_zero_for_orthogonal_queries():
    q = torch.eye(4, 16).unsqueeze(0) * 3.0
    assert head_torch.ortho_loss(q, 0.1).item() < 1e-12
    q2 = torch.ones(1, 2, 8)
    assert abs(head_torch.ortho_loss(q2, 0.1).item() - 0.1 * 2.0) < 1e-6   # ||[[0,1],[1,0]]||_F^2 = 2

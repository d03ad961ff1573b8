"""The training-side restatements against numbers the REFERENCE's own code produced
(tests/golden/make_golden_train.py executes /root/reference/main/losses/hash_loss.py:17-59 and the train-mode forwards of
/root/reference/main/models/multi_dino_attention.py:1001-1141, 336-599 in the build container; the fixture holds tensors
and key names only).  CPU tests: the training path of wvhash is stock PyTorch and follows its tensors' device."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from make_golden_train import HASHLOSS, TRAIN_HEAD_CASES, hashloss_inputs  # noqa: E402  (case tables and seeded inputs only)
from wvhash import synth  # noqa: E402
from wvhash.losses import HashLoss  # noqa: E402
from wvhash.models import get_fusion_head  # noqa: E402


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "train_golden.npz"))


def test_hashloss_is_the_references_loss(gold):
    """Same seed -> the same proxies (the reference draws randn before xavier_uniform_: the global RNG stream is consumed
    identically, checked by the next draws), the same loss value and gradients, the same proxies after one step of the
    loss's own AdamW, the same state_dict keys (the optimizer state travels inside, hash_loss.py:50-59)."""
    torch.manual_seed(HASHLOSS["seed"])
    loss = HashLoss(num_classes=HASHLOSS["num_classes"], embedding_size=HASHLOSS["embedding_size"])
    assert np.array_equal(torch.rand(4).numpy(), gold["hashloss/rng_after_init"])
    assert np.array_equal(loss.proxies.detach().numpy(), gold["hashloss/proxies_init"])
    emb, labels = hashloss_inputs()
    emb.requires_grad_(True)
    val = loss(emb, labels)
    val.backward()
    assert abs(val.item() - float(gold["hashloss/value"][0])) < 1e-6
    assert np.abs(emb.grad.numpy() - gold["hashloss/grad_embeddings"]).max() < 1e-7
    assert np.abs(loss.proxies.grad.numpy() - gold["hashloss/grad_proxies"]).max() < 1e-7
    loss.step()
    assert np.abs(loss.proxies.detach().numpy() - gold["hashloss/proxies_after_step"]).max() < 1e-7
    sd = loss.state_dict()
    assert sorted(sd.keys()) == list(gold["hashloss/state_dict_keys"])
    assert sorted(sd["optimizer_state"].keys()) == list(gold["hashloss/optimizer_state_keys"])
    # ... and it loads back strictly, optimizer moments included
    other = HashLoss(num_classes=HASHLOSS["num_classes"], embedding_size=HASHLOSS["embedding_size"])
    other.load_state_dict(sd, strict=True)
    assert torch.equal(other.proxies, loss.proxies)
    m0 = loss.loss_optimizer.state_dict()["state"][0]["exp_avg"]
    assert torch.equal(other.loss_optimizer.state_dict()["state"][0]["exp_avg"], m0)
    assert "optimizer_state" in sd                                   # load_state_dict did not eat the caller's dict


@pytest.mark.parametrize("case", TRAIN_HEAD_CASES, ids=[c[0] for c in TRAIN_HEAD_CASES])
def test_train_mode_heads_match_the_reference(gold, case):
    """.train() forward with dropout = 0 and sub_band_dropout_p = 0: output, last_ortho_loss (Advanced / Pooled / Decoupled:
    Gram of the raw query tokens, :1095-1109; the plain head: Gram of the batch-mean attention weights, :1047-1052) and the
    gradient reaching the query tokens through both."""
    name, ftype, nq, extra, B, seed = case
    E = 384
    cfg = {"type": ftype, "output_dim": E, "num_heads": 8, "dropout": 0.0, "num_queries": nq, "sub_band_dropout_p": 0.0,
           "ortho_weight": 0.1}
    cfg.update(extra)
    head = get_fusion_head(cfg, [E] * 4).train()
    pool = "mean" if extra.get("query_pool") == "mean" else "concat"
    head.load_state_dict(synth.head_state(E, nq, pool, seed, query_scale=extra.get("query_scale_init")), strict=True)
    feats = synth.band_features(B, E, seed + 1000)
    y = head([f.clone() for f in feats])
    ortho = head.last_ortho_loss
    (y.sum() + ortho).backward()
    assert np.abs(y.detach().numpy() - gold[f"{name}/train_out"]).max() < 1e-5
    assert abs(float(ortho.detach()) - float(gold[f"{name}/train_ortho"][0])) < 1e-6
    g_ref = gold[f"{name}/train_grad_query_tokens"]
    assert np.abs(head.query_tokens.grad.numpy() - g_ref).max() <= 1e-5 * max(1.0, np.abs(g_ref).max())

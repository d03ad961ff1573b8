#!/usr/bin/env python3
"""Regenerates tests/golden/train_golden.npz.  Run in the BUILD container only (needs /root/reference):
    python tests/golden/make_golden_train.py

What it pins -- the training-side restatements of SURVEY.md 8 f-3 -- with numbers the REFERENCE's own code produced here:

* `hashloss/*`: /root/reference/main/losses/hash_loss.py (imports torch only) is imported by file path; under a fixed global
  seed `HashLoss(num_classes, embedding_size)` is constructed (-> the proxies the reference's RNG consumption gives: randn,
  then xavier_uniform_), run on seeded embeddings / multi-hot labels (-> loss, gradients w.r.t. embeddings and proxies),
  stepped once with its own AdamW (-> proxies after the step), and its state_dict keys are listed.
* `<head>/train_*`: the reference's four cross-attention fusion heads (main/models/multi_dino_attention.py, imported by
  file path as in make_golden.py) in .train() mode with dropout = 0 and sub_band_dropout_p = 0 (no randomness left), on the
  seeded weights of wvhash.synth.head_state: output, last_ortho_loss, and the gradient of (output.sum() + ortho) w.r.t. the
  query tokens.

Fixtures are data only (tensors, key names).  tests/test_train_golden.py compares wvhash.losses.HashLoss and the
training-mode forward of wvhash.models.fusion to them on the CPU.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from make_golden import load_reference_heads  # noqa: E402
from wvhash import synth  # noqa: E402

REF = "/root/reference"

# name, fusion type, Nq, extra fusion_config, batch, seed
TRAIN_HEAD_CASES = [
    ("adv_nq4", "cross_attention_advanced", 4, {}, 6, 31),
    ("adv_nq8", "cross_attention_advanced", 8, {}, 4, 32),
    ("base_nq4", "cross_attention_bottleneck", 4, {}, 5, 33),
    ("pooled_nq4", "cross_attention_pooled", 4, {"query_pool": "mean"}, 5, 34),
    ("decoupled_nq4", "cross_attention_decoupled", 4, {"query_scale_init": 4.0, "normalize_queries": True}, 5, 35),
]
HASHLOSS = dict(num_classes=38, embedding_size=64, seed=1234, batch=24)


def hashloss_inputs():
    g = torch.Generator().manual_seed(77)
    emb = 1.5 * torch.randn(HASHLOSS["batch"], HASHLOSS["embedding_size"], generator=g)
    labels = synth.multi_hot_labels(HASHLOSS["batch"], HASHLOSS["num_classes"], 0.10, seed=78)
    return emb, labels


def load_reference_hashloss():
    spec = importlib.util.spec_from_file_location("ref_hash_loss", os.path.join(REF, "main", "losses", "hash_loss.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make():
    out = {}
    ref = load_reference_hashloss()
    torch.manual_seed(HASHLOSS["seed"])
    loss = ref.HashLoss(num_classes=HASHLOSS["num_classes"], embedding_size=HASHLOSS["embedding_size"])
    out["hashloss/rng_after_init"] = torch.rand(4).numpy()      # the global stream right after construction
    out["hashloss/proxies_init"] = loss.proxies.detach().clone().numpy()
    emb, labels = hashloss_inputs()
    emb.requires_grad_(True)
    val = loss(emb, labels)
    val.backward()
    out["hashloss/value"] = np.array([val.item()], dtype=np.float64)
    out["hashloss/grad_embeddings"] = emb.grad.numpy()
    out["hashloss/grad_proxies"] = loss.proxies.grad.detach().clone().numpy()
    loss.step()
    out["hashloss/proxies_after_step"] = loss.proxies.detach().clone().numpy()
    sd = loss.state_dict()
    out["hashloss/state_dict_keys"] = np.array(sorted(sd.keys()))
    out["hashloss/optimizer_state_keys"] = np.array(sorted(sd["optimizer_state"].keys()))
    print(f"HashLoss: value {val.item():.6f}, state_dict keys {sorted(sd.keys())}")

    mda = load_reference_heads()
    E, heads = 384, 8
    for name, ftype, nq, extra, B, seed in TRAIN_HEAD_CASES:
        cfg = {"type": ftype, "output_dim": E, "num_heads": heads, "dropout": 0.0, "num_queries": nq,
               "sub_band_dropout_p": 0.0, "ortho_weight": 0.1}
        cfg.update(extra)
        head = mda.get_fusion_head(cfg, [E] * 4).train()
        pool = "mean" if extra.get("query_pool") == "mean" else "concat"
        head.load_state_dict(synth.head_state(E, nq, pool, seed, query_scale=extra.get("query_scale_init")), strict=True)
        feats = synth.band_features(B, E, seed + 1000)
        y = head([f.clone() for f in feats])
        ortho = head.last_ortho_loss
        (y.sum() + ortho).backward()
        out[f"{name}/train_out"] = y.detach().numpy()
        out[f"{name}/train_ortho"] = np.array([float(ortho.detach())], dtype=np.float64)
        out[f"{name}/train_grad_query_tokens"] = head.query_tokens.grad.detach().clone().numpy()
        out[f"{name}/meta"] = np.array([nq, B, seed, 1 if pool == "mean" else 0], dtype=np.int64)
        print(f"train head {name}: out {tuple(y.shape)} ortho {float(ortho):.6f}")
    np.savez_compressed(os.path.join(HERE, "train_golden.npz"), **out)


if __name__ == "__main__":
    make()

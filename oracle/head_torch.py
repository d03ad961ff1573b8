"""oracle/head_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (explicit torch math, no nn.MultiheadAttention) of the eval-mode forward of
the band-attention pooling heads and of the hashing tail:

* ``band_attn_pool``  <- CrossAttentionBottleneckHeadAdvanced.forward
  /root/reference/main/models/multi_dino_attention.py:1111-1141 (same attention core in
  CrossAttentionBottleneckHead :1030-1062, ...Pooled :568-599 (mean read-out) and
  ...Decoupled :448-481 (q = scale * normalize(q), :436-446))
* ``ortho_loss``      <- compute_ortho_loss :1095-1109
* ``hash_tail``       <- SharedDinoHashing.forward tail :829-833 (hash_fc, BatchNorm1d in eval
  mode, sign)

PARITY STATUS: pinned.  tests/golden/head_*.pt hold inputs, state_dicts and outputs produced
by importing the reference module itself by file path (tests/golden/make_golden.py); this
restatement is checked against them in tests/test_oracle_head.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F


def effective_queries(sd, normalize_queries=False, prefix=""):
    q = sd[prefix + "query_tokens"]  # [1, Nq, E]
    if normalize_queries:
        q = F.normalize(q, p=2, dim=-1)
    if (prefix + "query_scale") in sd:
        q = q * sd[prefix + "query_scale"]
    return q


def band_attn_pool(features_list, sd, num_heads=8, pool="concat", normalize_queries=False,
                   prefix="", return_weights=False, dtype=None):
    """features_list: 4 x [B, E] (CLS feature per band LL, LH, HL, HH) -> [B, E]."""
    g = lambda k: sd[prefix + k] if dtype is None else sd[prefix + k].to(dtype)
    feats = [f if dtype is None else f.to(dtype) for f in features_list]
    kv = torch.stack(feats, dim=1)                                  # [B, S=4, E]
    B, S, E = kv.shape
    q = effective_queries(sd, normalize_queries, prefix)
    q = q if dtype is None else q.to(dtype)
    q = q.expand(B, -1, -1)                                         # [B, Nq, E]
    Nq = q.shape[1]
    hd = E // num_heads
    w_in, b_in = g("attn.in_proj_weight"), g("attn.in_proj_bias")
    Q = q @ w_in[:E].t() + b_in[:E]
    K = kv @ w_in[E:2 * E].t() + b_in[E:2 * E]
    V = kv @ w_in[2 * E:].t() + b_in[2 * E:]
    Qh = Q.view(B, Nq, num_heads, hd).transpose(1, 2)               # [B, h, Nq, hd]
    Kh = K.view(B, S, num_heads, hd).transpose(1, 2)
    Vh = V.view(B, S, num_heads, hd).transpose(1, 2)
    scores = (Qh @ Kh.transpose(-1, -2)) / math.sqrt(hd)            # [B, h, Nq, S]
    P = torch.softmax(scores, dim=-1)
    ctx = (P @ Vh).transpose(1, 2).reshape(B, Nq, E)
    attn_out = ctx @ g("attn.out_proj.weight").t() + g("attn.out_proj.bias")
    x = F.layer_norm(q + attn_out, (E,), g("norm1.weight"), g("norm1.bias"), 1e-5)
    h = x @ g("mlp.0.weight").t() + g("mlp.0.bias")
    h = F.gelu(h)                                                   # exact erf GELU
    x = x + (h @ g("mlp.2.weight").t() + g("mlp.2.bias"))
    x = x.mean(dim=1) if pool == "mean" else x.reshape(B, -1)
    x = x @ g("out_proj.weight").t() + g("out_proj.bias")
    out = F.layer_norm(x, (E,), g("norm2.weight"), g("norm2.bias"), 1e-5)
    if return_weights:
        return out, P.mean(dim=1)                                   # MHA averages heads
    return out


def ortho_loss(query_tokens, ortho_weight=0.1, margin=0.0):
    Q = query_tokens.squeeze(0)
    Qn = F.normalize(Q, p=2, dim=-1)
    gram = Qn @ Qn.T
    err = torch.norm(gram - torch.eye(Q.shape[0]), p="fro")
    return ortho_weight * (F.relu(err - margin) ** 2)


def hash_tail(fused, hash_w, bn_w, bn_b, bn_mean, bn_var, eps=1e-5, hash_b=None, return_logits=False):
    """hash_fc (no bias when BN is used) -> BatchNorm1d(eval, running stats) -> sign."""
    logits = fused @ hash_w.t()
    if hash_b is not None:
        logits = logits + hash_b
    if bn_w is not None:
        logits = (logits - bn_mean) / torch.sqrt(bn_var + eps) * bn_w + bn_b
    if return_logits:
        return logits
    return torch.sign(logits)

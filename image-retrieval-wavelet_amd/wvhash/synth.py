"""Seeded synthetic inputs of the shapes the reference's datasets / checkpoints would supply.

There are no datasets or checkpoints in the build or bench environments, so bench.py, smoke()
and the tests use these generators (definitions: SURVEY.md section 8d).  Pure data: nothing here
computes any part of the hot path.
"""
import hashlib

import numpy as np
import torch


# ------------------------------------------------------------------------------- images
def noise_images(batch, height=224, width=224, seed=1234):
    """White-noise RGB, uint8 [B, H, W, 3]."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (batch, height, width, 3), dtype=np.uint8)


def natural_images(batch, height=224, width=224, seed=1234, box=8):
    """Box-filtered noise rescaled to the full u8 range: detail std << LL std, like a photo."""
    rng = np.random.default_rng(seed)
    x = rng.random((batch, height + box, width + box, 3), dtype=np.float32)
    c = np.cumsum(np.cumsum(x, axis=1), axis=2)
    c = np.pad(c, ((0, 0), (1, 0), (1, 0), (0, 0)))
    s = c[:, box:, box:] - c[:, :-box, box:] - c[:, box:, :-box] + c[:, :-box, :-box]
    s = s[:, :height, :width]
    lo = s.min(axis=(1, 2, 3), keepdims=True)
    hi = s.max(axis=(1, 2, 3), keepdims=True)
    return np.round((s - lo) / (hi - lo) * 255.0).astype(np.uint8)


# ------------------------------------------------------------------------------- codes / labels
def random_codes(n_query, n_db, nbits, seed=0):
    """+-1 codes, queries first then database from one generator
    (call order of /root/reference/studies/measure_random_baseline.py:84,105-106)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randint(0, 2, (n_query, nbits), generator=g).float() * 2 - 1
    r = torch.randint(0, 2, (n_db, nbits), generator=g).float() * 2 - 1
    return q, r


def multi_hot_labels(n, n_classes, p, seed):
    """fp32 multi-hot labels, Bernoulli(p) per tag, every row has at least one tag."""
    g = torch.Generator().manual_seed(seed)
    lab = (torch.rand(n, n_classes, generator=g) < p).float()
    empty = lab.sum(1) == 0
    fill = torch.randint(0, n_classes, (n,), generator=g)
    lab[empty, fill[empty]] = 1.0
    return lab


def structured_codes(labels, nbits, w_seed, noise_seed, noise=0.5):
    """Label-correlated +-1 codes: sign(labels @ W + noise * N(0,1)); queries and database share
    `w_seed`."""
    W = torch.randn(labels.shape[1], nbits, generator=torch.Generator().manual_seed(w_seed))
    g = torch.Generator().manual_seed(noise_seed)
    z = labels @ W + noise * torch.randn(labels.shape[0], nbits, generator=g)
    c = torch.sign(z)
    c[c == 0] = 1.0
    return c


# ------------------------------------------------------------------------------- head weights
def head_state(embed_dim=384, num_queries=4, pool="concat", seed=0, query_scale=None):
    """Random, well-scaled state_dict with the key names of the reference's
    CrossAttentionBottleneckHead* modules (multi_dino_attention.py:1064-1093)."""
    g = torch.Generator().manual_seed(seed)
    E = embed_dim

    def w(out_f, in_f):
        return torch.randn(out_f, in_f, generator=g) / (in_f ** 0.5)

    def b(n):
        return 0.1 * torch.randn(n, generator=g)

    sd = {
        "query_tokens": 0.5 * torch.randn(1, num_queries, E, generator=g),
        "attn.in_proj_weight": w(3 * E, E),
        "attn.in_proj_bias": b(3 * E),
        "attn.out_proj.weight": w(E, E),
        "attn.out_proj.bias": b(E),
        "norm1.weight": 1.0 + 0.1 * torch.randn(E, generator=g),
        "norm1.bias": b(E),
        "norm2.weight": 1.0 + 0.1 * torch.randn(E, generator=g),
        "norm2.bias": b(E),
        "mlp.0.weight": w(4 * E, E),
        "mlp.0.bias": b(4 * E),
        "mlp.2.weight": w(E, 4 * E),
        "mlp.2.bias": b(E),
        "out_proj.weight": w(E, E if pool == "mean" else num_queries * E),
        "out_proj.bias": b(E),
    }
    if query_scale is not None:
        sd["query_scale"] = torch.tensor(float(query_scale))
    return sd


def hash_tail_state(embed_dim=384, nbits=64, seed=0):
    """hash_fc (no bias) + BatchNorm1d running statistics (SharedDinoHashing :810-813)."""
    g = torch.Generator().manual_seed(seed)
    return {
        "hash_fc.weight": torch.randn(nbits, embed_dim, generator=g) / (embed_dim ** 0.5),
        "bn.weight": 1.0 + 0.1 * torch.randn(nbits, generator=g),
        "bn.bias": 0.1 * torch.randn(nbits, generator=g),
        "bn.running_mean": 0.1 * torch.randn(nbits, generator=g),
        "bn.running_var": 0.5 + torch.rand(nbits, generator=g),
        "bn.num_batches_tracked": torch.tensor(1),
    }


def band_features(batch, embed_dim=384, seed=0):
    """4 x [B, E] CLS features (LL, LH, HL, HH)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, embed_dim, generator=g) for _ in range(4)]


def state_sha(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def randomize_module(module, seed):
    """Deterministic, well-scaled values for EVERY parameter and floating-point buffer of a module (sorted by name, one
    generator): the same call on a structurally identical module -- the reference's class with the same state_dict keys --
    gives the same numbers, so fixtures need to hold outputs only."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5))
            else:
                scale_like = name.endswith("weight")                 # 1-D weights are norm scales: around one
                p.copy_(0.1 * torch.randn(p.shape, generator=g) + (1.0 if scale_like else 0.0))
        for name, b in sorted(module.named_buffers()):
            if not b.dtype.is_floating_point:
                continue
            if name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            else:
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
    return module

"""The reference's non-cross-attention fusion heads (standard, temperature, semantic, gated, temperature_gated, self_attention,
cbam, eca) -- stock PyTorch modules outside the accelerated path, kept for the drop-in contract: get_fusion_head selects them
with the reference's config keys, their state_dict keys are the reference's (the fixture generator loads them into the
reference's classes with strict=True) and their outputs equal the reference modules' outputs (tests/golden/
fusion_extra_golden.npz, made by executing multi_dino_attention.py:156-334 in the build container).  Runs on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

from wvhash import synth
from wvhash.models import get_fusion_head
from wvhash.models import fusion_extra as FX

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fusion_extra_cases.json")) as _fh:
    EXTRA_HEAD_CASES = [(c["name"], c["config"], c["input_dims"], c["batch"], c["seed"]) for c in json.load(_fh)]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/fusion_extra_golden.npz")


@pytest.mark.parametrize("name,cfg,dims,B,seed", EXTRA_HEAD_CASES, ids=[c[0] for c in EXTRA_HEAD_CASES])
def test_outputs_equal_the_reference_modules(gold, name, cfg, dims, B, seed):
    head = synth.randomize_module(get_fusion_head(dict(cfg), list(dims)), seed).eval()
    g = torch.Generator().manual_seed(seed + 500)
    feats = [torch.randn(B, d, generator=g) for d in dims]
    with torch.no_grad():
        y = head(feats)
    want = gold[name + "/out"]
    assert tuple(y.shape) == want.shape
    assert np.abs(y.numpy() - want).max() < 2e-5, name


def test_dispatch_and_state_dict_keys():
    kinds = {"standard": FX.StandardFusionHead, "temperature": FX.TemperatureFusionHead, "semantic": FX.SemanticFusionHead,
             "gated": FX.GatedFusionHead, "temperature_gated": FX.TemperatureGatedFusionHead,
             "self_attention": FX.AttentionFusionHead, "cbam": FX.AdvancedFusionModule, "eca": FX.AdvancedFusionModule,
             "anything_else": FX.StandardFusionHead}
    for ftype, cls in kinds.items():
        head = get_fusion_head({"type": ftype, "output_dim": 64}, [64] * 4)
        assert type(head) is cls, ftype
    keys = set(get_fusion_head({"type": "cbam", "output_dim": 64}, [64] * 4).state_dict())
    assert {"gate.ChannelGate.mlp.1.weight", "gate.ChannelGate.mlp.3.bias", "fcn.0.weight", "fcn.1.running_mean"} <= keys
    assert "gate.conv.weight" in get_fusion_head({"type": "eca", "output_dim": 64}, [64] * 4).state_dict()
    tg = get_fusion_head({"type": "temperature_gated", "output_dim": 64, "temperature": 0.3}, [64] * 4)
    assert tg.temperature == 0.3 and len(tg.gate_network) == 3 and len(get_fusion_head({"type": "gated", "output_dim": 64}, [64] * 4).gate_network) == 4
    proj = get_fusion_head({"type": "standard", "output_dim": 32}, [48, 32, 16, 32])
    assert [type(p).__name__ for p in proj.projections] == ["Linear", "Identity", "Linear", "Identity"]


@pytest.mark.parametrize("ftype", ["standard", "gated", "eca"])
def test_training_mode_backward_reaches_every_parameter(ftype):
    torch.manual_seed(0)
    head = get_fusion_head({"type": ftype, "output_dim": 32, "num_heads": 4}, [32] * 4).train()
    feats = [torch.randn(6, 32, requires_grad=True) for _ in range(4)]
    head(feats).square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())
    assert all(f.grad is not None for f in feats)


def test_hashing_model_accepts_a_stock_head():
    from wvhash.models import SharedDinoHashing
    from wvhash.models.vit import tiny_vit
    net = SharedDinoHashing({"name": "dinov2_vits14"}, {"type": "gated", "output_dim": 384}, {"nbits": 16}, backbone=tiny_vit()).eval()
    assert isinstance(net.fusion_head, FX.GatedFusionHead)

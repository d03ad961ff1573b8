"""GPU parity of the HIP/MFMA band-attention head and the hashing tail against the outputs of the
REFERENCE's own modules (tests/golden/head_golden.npz) and against oracle/head_torch.py.

Tolerance: every contraction is fp32 (v_mfma_f32_32x32x2_f32 = fmaf chain); only the summation
order differs from ATen's blocked fp32 GEMM.  Outputs are LayerNorm'ed (|y| ~ 3): atol 5e-5.
"""
import numpy as np
import pytest
import torch

from oracle import head_torch
from wvhash import synth
from wvhash.models import get_fusion_head, hash_tail, SharedDinoHashing, MultiDinoHashing
from wvhash.models.vit import tiny_vit

pytestmark = pytest.mark.gpu

ATOL = 5e-5
TYPES = {"adv": "cross_attention_advanced", "base": "cross_attention_bottleneck",
         "pooled": "cross_attention_pooled", "decoupled": "cross_attention_decoupled"}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(f"{golden_dir}/head_golden.npz")


def build(n, gold):
    E, heads, nq, B, seed, mean, dec = gold[n + "/meta"].tolist()
    cfg = {"type": TYPES[n.split("_")[0]], "output_dim": E, "num_heads": heads, "num_queries": nq,
           "sub_band_dropout_p": 0.0, "ortho_weight": 0.1}
    if mean:
        cfg["query_pool"] = "mean"
    qs = float(gold[n + "/qscale"][0])
    if dec:
        cfg["query_scale_init"] = qs
    head = get_fusion_head(cfg, [E] * 4)
    sd = synth.head_state(E, nq, "mean" if mean else "concat", seed, query_scale=qs if dec else None)
    head.load_state_dict(sd)
    return head.cuda().eval(), synth.band_features(B, E, seed + 1000), sd


@pytest.mark.parametrize("front", ["0", "1"])
def test_heads_match_reference_module_outputs(gold, front, diag):
    """front = "1": the one-launch front (csrc/head_front.hip) wherever the configuration has one (E = 384 with 4 or 8
    queries: five of the seven cases); "0": the separate launches for all of them."""
    diag.setenv("WV_HEAD_FRONT", front)
    names = sorted({k.split("/")[0] for k in gold.files if k.endswith("/meta")})
    assert len(names) == 7
    for n in names:
        head, feats, _ = build(n, gold)
        with torch.no_grad():
            y = head([f.cuda() for f in feats])
        assert y.shape == gold[n + "/out"].shape
        assert np.abs(y.cpu().numpy() - gold[n + "/out"]).max() < ATOL, n
        assert float(head.last_ortho_loss) == 0.0


@pytest.mark.parametrize("front", ["0", "1"])
@pytest.mark.parametrize("B", [1, 63, 256, 2048])
def test_batch_sizes_against_oracle(B, front, diag):
    diag.setenv("WV_HEAD_FRONT", front)
    sd = synth.head_state(384, 4, "concat", seed=5)
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4}, [384] * 4)
    head.load_state_dict(sd)
    head = head.cuda().eval()
    feats = synth.band_features(B, 384, seed=B)
    with torch.no_grad():
        y = head([f.cuda() for f in feats])
        y2 = head([f.cuda() for f in feats])
    assert torch.equal(y, y2)                                   # deterministic
    ref = head_torch.band_attn_pool(feats, sd, 8)
    assert (y.cpu() - ref).abs().max().item() < ATOL
    ref64 = head_torch.band_attn_pool(feats[:1] if False else feats, sd, 8, dtype=torch.float64)
    assert (y.cpu().double() - ref64).abs().max().item() < ATOL


@pytest.mark.parametrize("nq,heads,B", [(4, 8, 2048), (4, 8, 1155), (8, 8, 600), (4, 12, 100), (8, 6, 37), (4, 16, 9),
                                        (8, 16, 70), (4, 2, 17)])
def test_one_launch_front_against_separate_launches_and_oracle(nq, heads, B, diag):
    """The fused front folds the K projection into the query tokens and mixes V in registers: same function, other
    summation order.  Both paths must sit within the golden tolerance of the fp64 oracle and within 2e-5 of each other;
    the prepared weight stream covers 1, 2 and 4 score blocks (Nq * heads = 32 ... 128) and partial last workgroups."""
    sd = synth.head_state(384, nq, "concat", seed=nq * 31 + heads)
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": nq, "num_heads": heads},
                           [384] * 4)
    head.load_state_dict(sd)
    head = head.cuda().eval()
    feats = synth.band_features(B, 384, seed=B + heads)
    dev = [f.cuda() for f in feats]
    out = {}
    with torch.no_grad():
        for front in ("1", "0"):
            diag.setenv("WV_HEAD_FRONT", front)
            out[front] = head(dev).cpu()
    assert head._qproj_cache["blob"] is not None                 # this configuration has a prepared stream
    assert not torch.equal(out["0"], out["1"])                   # two different kernels did run
    assert (out["0"] - out["1"]).abs().max().item() < 2e-5
    ref = head_torch.band_attn_pool(feats, sd, heads, dtype=torch.float64)
    for front in ("0", "1"):
        assert (out[front].double() - ref).abs().max().item() < ATOL, front


def test_front_is_chosen_by_batch_size_and_unsupported_shapes_fall_back(diag):
    diag.delenv("WV_HEAD_FRONT", raising=False)
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4}, [384] * 4)
    head.load_state_dict(synth.head_state(384, 4, "concat", seed=2))
    head = head.cuda().eval()
    small = [f.cuda() for f in synth.band_features(64, 384, seed=1)]
    big = [f.cuda() for f in synth.band_features(1536, 384, seed=1)]
    with torch.no_grad():
        auto_small, auto_big = head(small), head(big)
        diag.setenv("WV_HEAD_FRONT", "0")
        sep_small, sep_big = head(small), head(big)
    assert torch.equal(auto_small, sep_small)                    # 8 workgroups: the separate launches
    assert not torch.equal(auto_big, sep_big) and (auto_big - sep_big).abs().max().item() < 2e-5   # 192 workgroups: one launch
    diag.setenv("WV_HEAD_FRONT", "1")
    other = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 64, "num_queries": 4, "num_heads": 4}, [64] * 4)
    other = other.cuda().eval()                                   # E = 64 has no fused kernel: no blob, separate launches
    with torch.no_grad():
        y = other([f.cuda() for f in synth.band_features(5, 64, seed=3)])
    assert other._qproj_cache["blob"] is None and other._qproj_cache["qp"] is not None and torch.isfinite(y).all()


def test_hip_path_equals_stock_torch_forward_of_same_module():
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4}, [384] * 4)
    head.load_state_dict(synth.head_state(384, 4, "concat", seed=9))
    head = head.cuda().eval()
    feats = [f.cuda() for f in synth.band_features(32, 384, seed=10)]
    with torch.no_grad():
        y = head(feats)
    grad_feats = [f.clone().requires_grad_(True) for f in feats]   # autograd needed -> torch path
    y_t = head(grad_feats)
    assert y_t.requires_grad and (y - y_t.detach()).abs().max().item() < ATOL


def test_hash_tail_against_reference_and_packing(gold):
    tail = synth.hash_tail_state(384, 64, seed=21)
    fc = torch.nn.Linear(384, 64, bias=False)
    bn = torch.nn.BatchNorm1d(64)
    fc.load_state_dict({"weight": tail["hash_fc.weight"]})
    bn.load_state_dict({k[3:]: v for k, v in tail.items() if k.startswith("bn.")})
    fc, bn = fc.cuda().eval(), bn.cuda().eval()
    fused = torch.from_numpy(gold["adv_e384_nq4/out"]).cuda()
    out = hash_tail(fused, fc, bn, want=("logits", "codes", "packed"))
    np.testing.assert_allclose(out["logits"].cpu().numpy(), gold["tail/logits"], atol=5e-6)
    far = np.abs(gold["tail/logits"]) > 1e-4
    assert np.array_equal(out["codes"].cpu().numpy()[far], gold["tail/codes"][far])
    from wvhash.engine import hamming as H
    assert torch.equal(out["packed"], H.pack_codes(out["codes"]))
    # 128-bit codes, bias and no BN
    fc2 = torch.nn.Linear(384, 128, bias=True).cuda().eval()
    big = torch.randn(300, 384, device="cuda")
    o2 = hash_tail(big, fc2, torch.nn.Identity(), want=("logits", "codes", "packed"))
    ref = torch.nn.functional.linear(big.cpu(), fc2.weight.cpu(), fc2.bias.cpu())
    assert (o2["logits"].cpu() - ref).abs().max().item() < 1e-5
    assert torch.equal(o2["packed"], H.pack_codes(o2["codes"]))


def test_model_classes_end_to_end_with_stub_backbone():
    torch.manual_seed(0)
    fusion = {"type": "cross_attention_advanced", "output_dim": 384, "num_heads": 8, "num_queries": 4,
              "sub_band_dropout_p": 0, "ortho_weight": 0.1, "dropout": 0.1}
    net = SharedDinoHashing({"name": "dinov2_vits14", "frozen": True}, fusion, {"nbits": 64},
                            backbone=tiny_vit(), modelhooks={"name": "x"}, with_autocast=True).cuda().eval()
    net.set_wavelet(level=1, wavelet="haar")
    img = torch.from_numpy(synth.natural_images(4, 224, 224, seed=3)).permute(0, 3, 1, 2).contiguous().cuda()
    from wvhash.transforms import swt2d
    with torch.no_grad():
        codes_raw = net(img)                                     # 4-D raw batch -> SWT on device
        codes_5d = net(swt2d(img, "haar", 1))                    # reference-style 5-D input
        packed = net.encode_packed(img)
    assert tuple(codes_raw.shape) == (4, 64) and torch.equal(codes_raw, codes_5d)
    assert set(codes_raw.unique().tolist()) <= {-1.0, 0.0, 1.0}
    from wvhash.engine import hamming as H
    assert torch.equal(packed, H.pack_codes(codes_raw, check=False))
    # stock-torch evaluation of the same tail agrees except at |logit| ~ 0
    with torch.no_grad():
        fused = net.fused_embedding(img)
        logits = net.bn(net.hash_fc(fused))
    far = logits.abs() > 1e-4
    assert torch.equal(torch.sign(logits)[far], codes_raw[far])
    multi = MultiDinoHashing([{"name": "dinov2_vits14"}] * 4, fusion, {"nbits": 32},
                             backbones=[tiny_vit() for _ in range(4)]).cuda().eval()
    with torch.no_grad():
        assert tuple(multi(swt2d(img, "haar", 1)).shape) == (4, 32)


@pytest.mark.parametrize("B,nbits,with_bn", [(64, 64, True), (301, 64, True), (2048, 64, True), (130, 128, False), (77, 48, True)])
def test_hash_tail_kernels_agree_and_do_not_depend_on_the_batch(B, nbits, with_bn, diag):
    """Three kernels behind wv_hash_tail: per sample (fmaf chain), 16 samples per workgroup on the VALU (the same chain:
    bit-identical to it), and the matrix-core one (32 samples per workgroup, four k ranges summed in a fixed order: other
    rounding, within 1e-5 of the chain; codes equal wherever the logit is not within 1e-4 of zero).  Whatever the kernel,
    a sample's logits must not depend on the batch it arrives in."""
    torch.manual_seed(B + nbits)
    fc = torch.nn.Linear(384, nbits, bias=not with_bn).cuda().eval()
    bn = torch.nn.BatchNorm1d(nbits).cuda().eval() if with_bn else torch.nn.Identity()
    if with_bn:
        with torch.no_grad():
            bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
    x = torch.randn(B, 384, device="cuda")
    want = ("logits", "codes", "packed")
    out = {}
    for kern in ("simple", "valu16", "mfma"):
        diag.setenv("WV_HASH_TAIL", kern)
        out[kern] = hash_tail(x, fc, bn, want=want)
    for k in want:
        assert torch.equal(out["valu16"][k], out["simple"][k]), k
    assert (out["mfma"]["logits"] - out["simple"]["logits"]).abs().max().item() < 1e-5
    far = out["simple"]["logits"].abs() > 1e-4
    assert torch.equal(out["mfma"]["codes"][far], out["simple"]["codes"][far])
    from wvhash.engine import hamming as H
    assert torch.equal(out["mfma"]["packed"], H.pack_codes(out["mfma"]["codes"]))
    diag.delenv("WV_HASH_TAIL")
    auto = hash_tail(x, fc, bn, want=want)                       # what callers get: the matrix-core kernel from 32 samples on
    for k in want:
        assert torch.equal(auto[k], out["mfma"][k]), k
    diag.setenv("WV_HASH_TAIL", "mfma")
    for lo, hi in ((0, 1), (5, 37), (B - 33, B), (B // 2, B // 2 + 7)):
        part = hash_tail(x[lo:hi].contiguous(), fc, bn, want=want)
        for k in want:
            assert torch.equal(part[k], out["mfma"][k][lo:hi]), (k, lo, hi)


@pytest.mark.parametrize("ftype", ["cross_attention_advanced", "cross_attention_decoupled"])
@pytest.mark.parametrize("front", ["0", "1"])
def test_cached_query_projection_follows_parameter_updates(ftype, front, diag):
    """The projected query tokens are kept between calls (they are parameters); an in-place update of the query
    tokens, of the in-projection or of the query scale must invalidate them."""
    diag.setenv("WV_HEAD_FRONT", front)
    torch.manual_seed(3)
    head = get_fusion_head({"type": ftype, "output_dim": 384, "num_queries": 4, "sub_band_dropout_p": 0.0}, [384] * 4)
    head = head.cuda().eval()
    feats = [f.cuda() for f in synth.band_features(9, 384, seed=7)]

    def ref():
        sd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
        if ftype.endswith("decoupled"):
            q = torch.nn.functional.normalize(sd["query_tokens"], p=2, dim=-1) * sd.pop("query_scale")
            sd["query_tokens"] = q
        return head_torch.band_attn_pool([f.cpu() for f in feats], sd, 8)

    with torch.no_grad():
        y0 = head(feats)
        assert (y0.cpu() - ref()).abs().max() < ATOL
        key0 = head._qproj_cache["key"]
        head(feats)
        assert head._qproj_cache["key"] == key0                       # second call: projection reused
        head.query_tokens.mul_(1.5)
        y1 = head(feats)
        assert head._qproj_cache["key"] != key0 and (y1.cpu() - ref()).abs().max() < ATOL
        head.attn.in_proj_weight.add_(0.01)
        y2 = head(feats)
        assert (y2.cpu() - ref()).abs().max() < ATOL and not torch.equal(y1, y2)
        for w in (head.attn.out_proj.weight, head.mlp[0].weight, head.mlp[2].weight):   # copied into the prepared stream
            w.add_(0.01)
            y2b = head(feats)
            assert (y2b.cpu() - ref()).abs().max() < ATOL and not torch.equal(y2, y2b)
            y2 = y2b
        if ftype.endswith("decoupled"):
            head.query_scale.mul_(0.5)
            y3 = head(feats)
            assert (y3.cpu() - ref()).abs().max() < ATOL and not torch.equal(y2, y3)

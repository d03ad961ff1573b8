"""The training step of BASELINE configs c2 and c4 on the GPU: raw uint8 batches of a deferred
transform -> batched HIP SWT inside the step (bf16 sub-bands under bf16 autocast) -> stock-PyTorch forward/backward of
the band-attention hashing model with the proxy hashing loss + orthogonality loss -> optimizer step.
Reference: run.py:162-166, main/engine/base_update.py:52-292, main/losses/hash_loss.py:17-59."""
import pytest
import torch

from wvhash import synth
from wvhash.engine import backward_step, make_averager, train_step
from wvhash.losses import HashLoss
from wvhash.models import MultiDinoHashing, SharedDinoHashing
from wvhash.models.vit import tiny_vit
from wvhash.transforms import SWTTransform, swt2d

pytestmark = pytest.mark.gpu


def build(cls, nq, nbits, dropout=0.1, level=1, wavelet="haar"):
    torch.manual_seed(0)
    fusion = {"type": "cross_attention_advanced", "output_dim": 384, "num_heads": 8, "num_queries": nq, "dropout": dropout,
              "sub_band_dropout_p": 0.0, "ortho_weight": 0.1}
    if cls is MultiDinoHashing:
        net = cls([{"name": "dinov2_vits14", "frozen": True}] * 4, fusion, {"nbits": nbits}, backbones=[tiny_vit() for _ in range(4)])
    else:
        net = cls({"name": "dinov2_vits14", "frozen": True}, fusion, {"nbits": nbits}, backbone=tiny_vit())
    for name, p in net.named_parameters():
        if name.endswith(".gamma"):
            p.data.fill_(1.0)
    net = net.cuda().train().bind_transform(SWTTransform(level=level, wavelet=wavelet, defer=True))
    crit = HashLoss(num_classes=38, embedding_size=nbits).cuda()
    return net, [(crit, 1.0)]


def batch(n=16, seed=1):
    x = torch.from_numpy(synth.natural_images(n, 224, 224, seed=seed)).permute(0, 3, 1, 2).contiguous()   # raw uint8
    return x, synth.multi_hot_labels(n, 38, 0.10, seed)


def test_c2_shared_dino_training_steps_on_raw_batches():
    net, criteria = build(SharedDinoHashing, nq=4, nbits=64, level=3, wavelet="db2")
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    avg = make_averager(net, criteria)                            # world size 1: a no-op, same call as under torchrun
    x, y = batch()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    losses = []
    for _ in range(6):
        logs = train_step(net, x, y, criteria, {"net": opt}, averager=avg, autocast_dtype=torch.bfloat16, clip_grad=5.0)
        assert {"HashLoss", "Ortho_Loss", "total_loss"} <= set(logs) and all(torch.isfinite(torch.tensor(v)) for v in logs.values())
        losses.append(logs["HashLoss"])
    assert losses[-1] < losses[0]                                 # the same batch, six steps: the loss goes down
    after = net.state_dict()
    assert not torch.equal(before["fusion_head.query_tokens"], after["fusion_head.query_tokens"])
    assert not torch.equal(before["hash_fc.weight"], after["hash_fc.weight"])
    assert all(torch.equal(before[k], after[k]) for k in before if k.startswith("shared_backbone"))   # frozen
    assert all(p.grad is None for p in net.shared_backbone.parameters())


def test_gradient_cached_micro_batches_equal_the_single_pass_gradient():
    x, y = batch(12, seed=2)
    grads = []
    for sub in (None, 5):
        net, criteria = build(SharedDinoHashing, nq=4, nbits=64, dropout=0.0)
        net.bn.eval()                                             # the model's own BatchNorm sees micro-batch statistics otherwise
        backward_step(net, x, y, criteria, autocast_dtype=None, sub_batch=sub)
        grads.append({n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
    assert set(grads[0]) == set(grads[1]) and "fusion_head.mlp.0.weight" in grads[0]
    for k in grads[0]:
        assert torch.allclose(grads[0][k], grads[1][k], rtol=2e-3, atol=2e-6), k


def test_c4_multi_dino_bf16_step_feeds_bf16_sub_bands():
    net, criteria = build(MultiDinoHashing, nq=8, nbits=128)
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    x, y = batch(8, seed=3)
    seen = []
    hooks = [b.register_forward_pre_hook(lambda mod, args: seen.append((args[0].dtype, tuple(args[0].shape)))) for b in net.backbones]
    logs = train_step(net, x, y, criteria, [opt], autocast_dtype=torch.bfloat16, sub_batch=4)
    for h in hooks:
        h.remove()
    assert all(dt == torch.bfloat16 and shp[1:] == (3, 224, 224) for dt, shp in seen)   # the kernel wrote bf16 bands
    assert len(seen) == 4 * 2 * 2                                 # 4 backbones x 2 micro-batches x (no-grad pass + replay)
    assert torch.isfinite(torch.tensor(logs["total_loss"]))
    with torch.no_grad():                                          # eval after the step: the HIP head + tail take over
        codes = net.eval()(swt2d(x.cuda(), "haar", 1))
    assert tuple(codes.shape) == (8, 128) and set(codes.unique().tolist()) <= {-1.0, 1.0}

"""Training-side data parallelism (wvhash/engine/train_step.py) on the CPU with gloo, world size 2:
the averaged gradients of two ranks, each on its half of the batch, equal what ONE process gets when it runs the two
halves one after the other and averages -- for the single-pass step and for gradient-cached micro-batching -- and the
parameters after an optimizer step are identical on both ranks.  (The reference's nn.DataParallel, run.py:162-166, also
gives every replica its own slice and its own BatchNorm statistics.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed=0):
    from wvhash.losses import HashLoss
    from wvhash.models import SharedDinoHashing
    from wvhash.models.vit import ViT
    torch.manual_seed(seed)
    fusion = {"type": "cross_attention_advanced", "output_dim": 48, "num_heads": 4, "num_queries": 4, "dropout": 0.0,
              "sub_band_dropout_p": 0.0, "ortho_weight": 0.1}
    net = SharedDinoHashing({"name": "dinov2_vits14", "frozen": True}, fusion, {"nbits": 16},
                            backbone=ViT(48, 1, 4, 14, 28))
    for name, p in net.named_parameters():
        if name.endswith(".gamma"):
            p.data.fill_(1.0)
    crit = HashLoss(num_classes=6, embedding_size=16)
    return net.train(), [(crit, 1.0)]


def _data():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(12, 3, 4, 28, 28, generator=g)
    y = (torch.rand(12, 6, generator=g) < 0.3).float()
    return x, y


def _grads(net, criteria):
    named = list(net.named_parameters()) + [("proxies", criteria[0][0].proxies)]
    return {n: p.grad.clone() for n, p in named if p.requires_grad and p.grad is not None}


def _worker(rank, world, port, sub_batch, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash.engine import backward_step, make_averager, train_step
    net, criteria = _build()
    x, y = _data()
    n = x.shape[0] // world
    xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
    avg = make_averager(net, criteria, bucket_mb=0.01)          # tiny buckets: several all-reduces in flight
    assert len(avg.buckets) > 2
    backward_step(net, xs, ys, criteria, autocast_dtype=None, sub_batch=sub_batch)
    avg.average()
    torch.save(_grads(net, criteria), os.path.join(out_dir, f"g{rank}.pt"))
    net.zero_grad(); criteria[0][0].zero_grad()
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    train_step(net, xs, ys, criteria, [opt], averager=avg, autocast_dtype=None, sub_batch=sub_batch, clip_grad=1.0)
    torch.save({k: v.clone() for k, v in net.state_dict().items() if "backbone" not in k}, os.path.join(out_dir, f"p{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sub_batch", [None, 4])
def test_two_rank_gradients_equal_the_single_process_average(tmp_path, sub_batch):
    from wvhash.engine import backward_step
    port = 29600 + (os.getpid() + (sub_batch or 0) * 13) % 2000
    mp.spawn(_worker, args=(2, port, sub_batch, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(2)]
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k]), k                      # both ranks hold the same averaged gradient
    # one process, the two halves one after the other (fresh model each time, like two replicas), then the mean
    x, y = _data()
    halves = []
    for r in range(2):
        net, criteria = _build()
        backward_step(net, x[r * 6:(r + 1) * 6], y[r * 6:(r + 1) * 6], criteria, autocast_dtype=None, sub_batch=sub_batch)
        halves.append(_grads(net, criteria))
    assert set(got[0]) == set(halves[0]) and "fusion_head.query_tokens" in got[0] and "proxies" in got[0]
    assert not any(k.startswith("shared_backbone") for k in got[0])        # the frozen backbone is not exchanged
    for k in got[0]:
        want = 0.5 * (halves[0][k] + halves[1][k])
        assert torch.allclose(got[0][k], want, rtol=1e-5, atol=1e-7), k
    p0, p1 = (torch.load(os.path.join(tmp_path, f"p{r}.pt")) for r in range(2))
    for k in p0:
        if "num_batches_tracked" in k or "running_" in k:
            continue                                                       # BatchNorm statistics are per replica
        assert torch.equal(p0[k], p1[k]), k


def test_gradient_cached_step_equals_single_pass_on_one_process():
    """sub_batch micro-batching reproduces the full-batch gradient of everything that does not see batch statistics
    (base_update.py:151-292); the model's own BatchNorm1d sees micro-batch statistics, so it is put in eval mode here."""
    from wvhash.engine import backward_step
    x, y = _data()
    out = []
    for sub in (None, 5):
        net, criteria = _build()
        net.bn.eval()
        backward_step(net, x, y, criteria, autocast_dtype=None, sub_batch=sub)
        out.append(_grads(net, criteria))
    for k in out[0]:
        assert torch.allclose(out[0][k], out[1][k], rtol=2e-4, atol=1e-6), k

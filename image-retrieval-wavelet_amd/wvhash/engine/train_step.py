"""Training-side data parallelism for the wavelet-hashing models (BASELINE configs c2 and c4).

The reference trains with single-process ``nn.DataParallel`` (/root/reference/run.py:162-166): the model is replicated
onto every GPU each step, the batch is scattered, the outputs gathered on GPU 0, the loss computed there.  Here: one
process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI), every rank runs the whole step on its own
slice of the batch and the gradients are averaged with one bucketed all-reduce -- the optimisation step itself follows
the reference's ``_single_pass_optimization`` / ``_gradient_cached_optimization`` (main/engine/base_update.py:52-292):

* mixed precision by ``torch.autocast`` (bf16 on MI355X: no GradScaler needed; a scaler is honoured when given, like
  the reference's fp16 path);
* gradient-cached micro-batching (``sub_batch``): embeddings of all micro-batches without graph, the loss on the whole
  local batch, backward to the embeddings, then every micro-batch replayed with its RNG state and back-propagated with
  its slice of the cached gradient -- exactly the full-batch gradient at the memory of one micro-batch;
* the fusion head's orthogonality loss is picked up from ``fusion_head.last_ortho_loss`` (base_update.py:130-137);
* raw ``uint8 [B,3,H,W]`` batches of a deferred transform go straight into the model: the batched HIP SWT runs inside
  the step (bf16 sub-bands under bf16 autocast), so the wavelet expansion no longer happens per image on the host.

What is frozen stays out of the exchange: only parameters with ``requires_grad`` are bucketed (the DINOv2 backbones of
the headline configs are frozen, so a step moves the head, ``hash_fc``, ``bn`` -- a few MB -- over xGMI).
Training is outside the accelerated hot path: the forward/backward below is stock PyTorch on the GPU.
"""
import contextlib
import logging

import torch
import torch.distributed as dist

LOGGER = logging.getLogger("RETRIEVAL")


class GradientAverager(object):
    """Averages the gradients of ``parameters`` over the ranks of ``group`` with bucketed all-reduces.

    Buckets are flat fp32 buffers of about ``bucket_mb``: xGMI is point-to-point (7 links per GPU), a ring all-reduce is
    bound by one link, so few large messages beat many small ones; 32 MB keeps the latency term below 1 %.  All buckets
    are launched asynchronously and awaited together.  Parameters without a gradient in this step contribute zeros (every
    rank must bring the same buckets)."""

    def __init__(self, parameters, group=None, bucket_mb=32.0):
        self.params = [p for p in parameters if p.requires_grad]
        self.group = group
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets, cur, n = [], [], 0
        for p in self.params:
            if cur and n + p.numel() > limit:
                self.buckets.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += p.numel()
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def average(self):
        """In place: p.grad <- mean over ranks of p.grad, for every bucketed parameter."""
        world = self.world
        if world == 1:
            return
        works = []
        staged = dist.get_backend(self.group) == "gloo"             # gloo moves host memory (CPU rehearsal)
        for bi, bucket in enumerate(self.buckets):
            dev = bucket[0].device
            n = sum(p.numel() for p in bucket)
            flat = self._flat[bi]
            if flat is None or flat.numel() != n or flat.device != dev:
                flat = self._flat[bi] = torch.empty(n, dtype=torch.float32, device=dev)
            off = 0
            for p in bucket:
                seg = flat[off:off + p.numel()]
                if p.grad is None:
                    seg.zero_()
                else:
                    seg.copy_(p.grad.reshape(-1))
                off += p.numel()
            wire = flat.cpu() if (staged and flat.is_cuda) else flat
            works.append((bucket, flat, wire, dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        for bucket, flat, wire, work in works:
            work.wait()
            if wire is not flat:
                flat.copy_(wire)
            flat.mul_(1.0 / world)
            off = 0
            for p in bucket:
                seg = flat[off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = seg.clone().to(p.dtype)
                else:
                    p.grad.copy_(seg)
                off += p.numel()


def _autocast(device_type, dtype):
    if dtype is None:
        return contextlib.nullcontext()
    return torch.autocast(device_type, dtype=dtype)


def _rng_snapshot(device):
    return torch.get_rng_state(), (torch.cuda.get_rng_state(device) if device.type == "cuda" else None)


@contextlib.contextmanager
def _rng_replay(state, device):
    """Re-installs a captured RNG state for the duration of the block (dropout masks and the CPU coin of the LL-token
    masking must repeat when a micro-batch is forwarded a second time)."""
    cpu, gpu = state
    keep = _rng_snapshot(device)
    torch.set_rng_state(cpu)
    if gpu is not None:
        torch.cuda.set_rng_state(gpu, device)
    try:
        yield
    finally:
        torch.set_rng_state(keep[0])
        if keep[1] is not None:
            torch.cuda.set_rng_state(keep[1], device)


def _micro_batches(total, sub_batch):
    """[start, end) pairs; a trailing micro-batch of ONE sample is merged into its predecessor (BatchNorm in train
    mode refuses a batch of 1 -- same rule as base_update.py:35-53)."""
    starts = list(range(0, total, sub_batch))
    if len(starts) > 1 and total - starts[-1] == 1:
        starts.pop()
    return [(s, starts[i + 1] if i + 1 < len(starts) else total) for i, s in enumerate(starts)]


def _criterion_loss(criteria, emb, labels, logs):
    losses = []
    for crit, weight in criteria:
        if getattr(crit, "takes_embeddings", False):
            flat = labels.ndim == 1 or (labels.ndim == 2 and labels.size(1) == 1)
            loss = crit(emb, labels.view(-1) if flat else labels)
        else:   # pairwise losses see the score matrix and the label matrix (base_update.py:96-98)
            same = labels.view(-1, 1) == labels.view(1, -1) if labels.ndim == 1 else (labels.float() @ labels.float().t()) > 0
            loss = crit(emb @ emb.t(), same.float())
        loss = loss.mean()
        losses.append(weight * loss)
        logs[type(crit).__name__] = float(loss.detach())
    return losses


def _ortho_loss(net):
    inner = getattr(net, "module", net)
    head = getattr(inner, "fusion_head", None)
    loss = getattr(head, "last_ortho_loss", None)
    if isinstance(loss, torch.Tensor) and loss.requires_grad:
        return loss
    return None


def backward_step(net, images, labels, criteria, autocast_dtype=torch.bfloat16, scaler=None, sub_batch=None):
    """Forward + backward of one LOCAL batch (gradients accumulate into .grad; no optimizer step, no exchange).
    images: what the DataLoader collated -- expanded sub-bands [B,3,4,H,W] or raw uint8 [B,3,H,W] (deferred transform
    bound to the model).  criteria: [(loss module, weight)].  Returns a dict of logged scalars."""
    device = next(p for p in net.parameters()).device
    images = images.to(device, non_blocking=True)
    labels = labels.to(device, non_blocking=True)
    total = images.shape[0]
    logs = {}

    def scaled(t):
        return scaler.scale(t) if scaler is not None else t

    if not sub_batch or sub_batch >= total:
        with _autocast(device.type, autocast_dtype):
            emb = net(images)
            losses = _criterion_loss(criteria, emb, labels, logs)
        ortho = _ortho_loss(net)
        if ortho is not None:
            losses.append(ortho)
            logs["Ortho_Loss"] = float(ortho.detach())
        total_loss = sum(losses)
        scaled(total_loss).backward()
        logs["total_loss"] = float(total_loss.detach())
        return logs

    if sub_batch < 2:
        raise ValueError("sub_batch must be >= 2 (BatchNorm in train mode refuses a batch of one sample)")
    chunks = _micro_batches(total, sub_batch)
    # pass 1: embeddings of the whole local batch, no graph kept
    cached, states = [], []
    with torch.no_grad():
        for s, e in chunks:
            states.append(_rng_snapshot(device))
            with _autocast(device.type, autocast_dtype):
                cached.append(net(images[s:e]))
    emb_full = torch.cat(cached, dim=0).float().detach().requires_grad_()
    with _autocast(device.type, autocast_dtype):
        losses = _criterion_loss(criteria, emb_full, labels, logs)
        total_loss = sum(losses)
    logs["total_loss"] = float(total_loss.detach())
    scaled(total_loss).backward()                       # stops at the embeddings: one gradient row per sample
    # pass 2: replay every micro-batch with its graph and push its slice of the cached gradient through it
    for (s, e), state in zip(chunks, states):
        with _rng_replay(state, device):
            with _autocast(device.type, autocast_dtype):
                emb = net(images[s:e])
        tensors, grads = [emb], [emb_full.grad[s:e].to(emb.dtype)]
        ortho = _ortho_loss(net)
        if ortho is not None:                           # parameter-only loss: weighted by the chunk's share of the batch
            tensors.append(scaled(ortho * ((e - s) / total)))
            grads.append(torch.ones_like(ortho))
            if s == 0:
                logs["Ortho_Loss"] = float(ortho.detach())
        torch.autograd.backward(tensors, grad_tensors=grads)
    return logs


def train_step(net, images, labels, criteria, optimizers, averager=None, autocast_dtype=torch.bfloat16, scaler=None,
               sub_batch=None, clip_grad=None):
    """One optimisation step of one rank: backward_step on the local batch, gradient average over the ranks, optional
    clipping, optimizer steps (the model's optimizers and the losses' own, base_update.py:363-396), zero_grad."""
    logs = backward_step(net, images, labels, criteria, autocast_dtype=autocast_dtype, scaler=scaler, sub_batch=sub_batch)
    if averager is not None:
        averager.average()
    optimizers = list(optimizers.values()) if isinstance(optimizers, dict) else list(optimizers)
    if clip_grad is not None and clip_grad > 0.0:
        if scaler is not None:
            for opt in optimizers:
                scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=clip_grad)
    for opt in optimizers:
        scaler.step(opt) if scaler is not None else opt.step()
    for crit, _ in criteria:
        own = getattr(crit, "loss_optimizer", None)
        if own is not None:
            scaler.step(own) if scaler is not None else own.step()
            own.zero_grad()
    net.zero_grad()
    for crit, _ in criteria:
        crit.zero_grad()
    if scaler is not None:
        scaler.update()
    return logs


def make_averager(net, criteria=(), group=None, bucket_mb=32.0):
    """Averager over the trainable parameters of the model AND of the losses (HashLoss owns its proxies)."""
    params = [p for p in net.parameters() if p.requires_grad]
    for crit, _ in criteria:
        params += [p for p in crit.parameters() if p.requires_grad]
    return GradientAverager(params, group=group, bucket_mb=bucket_mb)

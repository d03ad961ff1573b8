"""The proxy hashing loss the reference's headline runs train with, so that the training step of configs c2 / c4
 can be exercised: same name, kwargs, parameters and numbers as
/root/reference/main/losses/hash_loss.py:17-59 (class-proxy BCE on scaled cosine similarities of tanh'ed logits +
quantisation pull towards +-1; the proxies have their own optimizer, stepped by the training loop).
Plain PyTorch: training is outside the accelerated path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class HashLoss(nn.Module):
    takes_embeddings = True

    def __init__(self, num_classes=20, embedding_size=64, quant_weight=0.1, scale=15.0, **kwargs):
        super().__init__()
        self.quant_weight, self.scale = quant_weight, scale
        # randn first, like the reference (:25-26): the values are overwritten, but the global RNG stream must be consumed the
        # same way or everything seeded after the loss differs from a reference run with the same seed
        self.proxies = nn.Parameter(torch.randn(num_classes, embedding_size))
        nn.init.xavier_uniform_(self.proxies)
        cfg = kwargs.get("optimizer") or {"name": "AdamW", "kwargs": {"lr": 1e-4, "weight_decay": 1e-4}}
        name = cfg.get("name", "AdamW") if isinstance(cfg, dict) else getattr(cfg, "name", "AdamW")
        kw = (cfg.get("kwargs", {}) if isinstance(cfg, dict) else getattr(cfg, "kwargs", {})) or {}
        self.loss_optimizer = getattr(torch.optim, name)(self.parameters(), **dict(kw))

    def forward(self, embeddings, labels, **kwargs):
        codes = torch.tanh(embeddings)
        sims = F.normalize(codes, p=2, dim=1) @ F.normalize(self.proxies, p=2, dim=1).t()
        bce = F.binary_cross_entropy_with_logits(sims * self.scale, labels.float())
        quant = (codes.abs() - 1.0).abs().mean()
        return bce + self.quant_weight * quant

    def step(self):
        self.loss_optimizer.step()
        self.loss_optimizer.zero_grad()

    # the proxies' optimizer travels inside the loss's state_dict (hash_loss.py:50-59): a reference-made checkpoint's
    # loss state loads strictly, and the moments survive a resume
    def state_dict(self, destination=None, prefix="", keep_vars=False):
        sd = super().state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)
        sd["optimizer_state"] = self.loss_optimizer.state_dict()
        return sd

    def load_state_dict(self, state_dict, strict=True):
        state_dict = dict(state_dict)
        optimizer_state = state_dict.pop("optimizer_state", None)
        out = super().load_state_dict(state_dict, strict)
        if optimizer_state is not None:
            self.loss_optimizer.load_state_dict(optimizer_state)
        return out

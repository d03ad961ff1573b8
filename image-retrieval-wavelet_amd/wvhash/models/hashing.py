"""Wavelet-band hashing models: class names, kwargs and state_dict keys of
/root/reference/main/models/multi_dino_attention.py (SharedDinoHashing :792-833,
MultiDinoHashing :716-750).

Only the tail of these models is on the accelerated path: band split -> [ViT] ->
fusion head (HIP/MFMA) -> hash_fc -> BatchNorm1d -> sign -> bit-pack (HIP).  The DINOv2 backbone
is a stock PyTorch module supplied by the caller (``backbone=`` / ``backbones=``) or fetched by
``load_dinov2`` (torch.hub; unavailable offline).  A 4-D input ``[B, 3, H, W]`` (raw image batch,
uint8 or float) is first expanded on the GPU by the batched SWT kernel, so the model also accepts
what a deferred ``SWTTransform`` emits; the reference's 5-D ``[B, 3, 4, H, W]`` input is accepted
unchanged.
"""
import torch
import torch.nn as nn

from .. import _lib
from ..transforms import functional as TF
from .fusion import get_fusion_head


def load_dinov2(name, retries=1, **kwargs):
    """The reference's backbone loader (main/models/hub_utils.py:35-59): torch.hub, i.e. network or a populated hub
    cache.  Offline this raises; pass ``backbone=`` / ``backbones=`` to the model classes instead (the backbone is
    outside the accelerated path; bench and tests use wvhash.models.vit)."""
    last = None
    for _ in range(max(1, retries)):
        try:
            return torch.hub.load("facebookresearch/dinov2:main", name, skip_validation=True, **kwargs)
        except Exception as e:  # network / cache miss
            last = e
    raise RuntimeError(f"load_dinov2('{name}') failed ({type(last).__name__}: {last}): construct the model with "
                       "backbone=<nn.Module with .embed_dim> instead")


def hash_tail(fused, hash_fc, bn, want=("codes",), host_twin=False):
    """HIP tail: logits = hash_fc(fused); bn (eval); sign; pack.  -> dict of requested outputs.
    host_twin=True (explicit; never a silent fallback): host tensors take the library's host twin wv_hash_tail_cpu."""
    host = bool(host_twin) and not fused.is_cuda
    lib = _lib.load() if host else _lib.require_gpu()
    fused = fused.float().contiguous()
    B, E = fused.shape
    nbits = hash_fc.out_features
    dev = fused.device
    out = {}
    logits = torch.empty((B, nbits), dtype=torch.float32, device=dev) if "logits" in want else None
    codes = torch.empty((B, nbits), dtype=torch.float32, device=dev) if "codes" in want else None
    packed = torch.empty((B, (nbits + 63) // 64), dtype=torch.int64, device=dev) if "packed" in want else None
    use_bn = isinstance(bn, nn.BatchNorm1d)
    w = hash_fc.weight.detach().float().contiguous()
    hb = hash_fc.bias.detach().float().contiguous() if hash_fc.bias is not None else None
    bw = bn.weight.detach().float().contiguous() if use_bn else None
    bb = bn.bias.detach().float().contiguous() if use_bn else None
    bm = bn.running_mean.detach().float().contiguous() if use_bn else None
    bv = bn.running_var.detach().float().contiguous() if use_bn else None
    if B and host:
        rc = lib.wv_hash_tail_cpu(_lib.ptr(fused), B, E, _lib.ptr(w), _lib.ptr(hb), _lib.ptr(bw), _lib.ptr(bb), _lib.ptr(bm),
                                  _lib.ptr(bv), float(bn.eps) if use_bn else 0.0, nbits, _lib.ptr(logits), _lib.ptr(codes),
                                  _lib.ptr(packed))
        _lib.check(rc, "wv_hash_tail_cpu")
    elif B:
        with torch.cuda.device(dev):
            rc = lib.wv_hash_tail(_lib.ptr(fused), B, E, _lib.ptr(w), _lib.ptr(hb), _lib.ptr(bw), _lib.ptr(bb),
                                  _lib.ptr(bm), _lib.ptr(bv), float(bn.eps) if use_bn else 0.0, nbits,
                                  _lib.ptr(logits), _lib.ptr(codes), _lib.ptr(packed), _lib.stream_ptr())
            _lib.check(rc, "wv_hash_tail")
    out["logits"], out["codes"], out["packed"] = logits, codes, packed
    return out


def _cls(out):
    return out['x_norm_clstoken'] if isinstance(out, dict) else out


class _WaveletHashingBase(nn.Module):
    """Shared input handling.  The reference's models take the expanded ``[B, 3, 4, H, W]`` sub-band tensor
    (multi_dino_attention.py:815-818, :745); these take that unchanged, and also the raw ``[B, 3, H, W]`` image
    batch a deferred transform emits -- expanded here by the batched HIP kernel of the transform that was BOUND to
    the model (``bind_transform``; ``engine.evaluate`` binds ``dataset.transform`` by itself).  A raw batch carries
    no wavelet name, so without a bound transform a 4-D input is an error, never a default wavelet."""

    _bound_transform = None

    def bind_transform(self, transform):
        """``transform``: the wavelet plugin (or a pipeline / dataset holding one) that produced the raw batches."""
        from ..transforms.custom_transforms import find_wavelet_transform
        found = find_wavelet_transform(transform)
        if found is None:
            raise ValueError("bind_transform: no SWTTransform / RawStackTransform / DWTTransform in the given pipeline")
        self._bound_transform = found
        return self

    def set_wavelet(self, level=1, wavelet="haar"):
        """Shorthand: raw batches are to be expanded by SWTTransform(level, wavelet)."""
        from ..transforms import SWTTransform
        return self.bind_transform(SWTTransform(level=level, wavelet=wavelet, defer=True))

    def _band_major(self, x):
        """-> [4, B, 3, H, W] (band-major; a view whenever the memory already lies that way)."""
        if x.dim() == 5:
            return x.permute(2, 0, 1, 3, 4)
        if x.dim() != 4:
            raise ValueError(f"expected [B,3,4,H,W] sub-bands or a [B,3,H,W] image batch, got {tuple(x.shape)}")
        tf = self._bound_transform
        if tf is None:
            raise RuntimeError(
                "this model received a raw [B,3,H,W] image batch (a deferred transform's output) but no transform "
                "is bound to it: call model.bind_transform(dataset.transform) (wvhash.engine.evaluate does) or "
                "model.set_wavelet(level, wavelet); the wavelet is never guessed")
        from ..transforms import SWTTransform
        if isinstance(tf, SWTTransform):
            # written band-major by the kernel: the band split below is a view, the sub-bands exist once in HBM.
            # Under bf16 autocast the backbone's first op casts its input to bf16 anyway: emit bf16 directly.
            bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            return TF.swt2d(x, tf.wavelet, tf.level, band_major=True,
                            out_dtype=torch.bfloat16 if bf16 else torch.float32)
        return tf.apply_batch(x).permute(2, 0, 1, 3, 4)

    def _tail(self, fused_embedding):
        host = not fused_embedding.is_cuda and getattr(self, "host_twin", False)      # explicit: model.host_twin = True
        if not self.training and (fused_embedding.is_cuda or host) and not torch.is_grad_enabled():
            return hash_tail(fused_embedding, self.hash_fc, self.bn, want=("codes",), host_twin=host)["codes"]
        logits = self.bn(self.hash_fc(fused_embedding))
        return self._train_output(logits) if self.training else torch.sign(logits)

    def encode_packed(self, x):
        """Eval-mode forward that keeps the codes on the GPU as packed int64 words."""
        fused = self.fused_embedding(x)
        return hash_tail(fused, self.hash_fc, self.bn, want=("packed",))["packed"]


class SharedDinoHashing(_WaveletHashingBase):
    def __init__(self, backbone_config, fusion_config, binary_config, backbone=None, **kwargs):
        super().__init__()
        self.shared_backbone = backbone if backbone is not None else load_dinov2(backbone_config['name'])
        if backbone_config.get('frozen', True):
            for p in self.shared_backbone.parameters():
                p.requires_grad = False
            self.shared_backbone.eval()
            self.shared_backbone.train = lambda mode=False: None
        embed_dim = self.shared_backbone.embed_dim
        self.fusion_head = get_fusion_head(fusion_config, [embed_dim] * 4)
        self.nbits = binary_config['nbits']
        self.bn = nn.BatchNorm1d(self.nbits)
        self.hash_fc = nn.Linear(fusion_config['output_dim'], self.nbits, bias=False)
        nn.init.normal_(self.hash_fc.weight, std=0.01)

    def _train_output(self, logits):
        return torch.tanh(logits)

    def fused_embedding(self, x):
        bands = self._band_major(x)                      # [4, B, 3, H, W]
        s, b, c, h, w = bands.shape
        x_concat = bands.reshape(s * b, c, h, w)         # a view for kernel-written bands; the reference's copy
        cls_tokens = _cls(self.shared_backbone(x_concat))  # (multi_dino_attention.py:818) for a 5-D input
        return self.fusion_head(list(cls_tokens.chunk(4, dim=0)))

    def forward(self, x):
        return self._tail(self.fused_embedding(x))


class MultiDinoHashing(_WaveletHashingBase):
    def __init__(self, backbones_config, fusion_config, binary_config, use_bn=True, backbones=None, **kwargs):
        super().__init__()
        self.backbones = nn.ModuleList()
        output_dims = []
        for i, bb_cfg in enumerate(backbones_config):
            model = backbones[i] if backbones is not None else load_dinov2(bb_cfg['name'])
            if bb_cfg.get('frozen', True):
                for p in model.parameters():
                    p.requires_grad = False
                model.eval()
                model.train = lambda mode=False: None
            self.backbones.append(model)
            output_dims.append(model.embed_dim)
        self.fusion_head = get_fusion_head(fusion_config, output_dims)
        self.nbits = binary_config['nbits']
        self.use_bn = use_bn
        self.bn = nn.BatchNorm1d(self.nbits) if use_bn else nn.Identity()
        self.hash_fc = nn.Linear(fusion_config['output_dim'], self.nbits, bias=not use_bn)
        nn.init.normal_(self.hash_fc.weight, std=0.01)
        if not use_bn:
            nn.init.zeros_(self.hash_fc.bias)

    def _train_output(self, logits):
        return logits

    def fused_embedding(self, x):
        bands = self._band_major(x)                      # bands[i] = x[..., i, :, :] (:745)
        features = [_cls(backbone(bands[i])) for i, backbone in enumerate(self.backbones)]
        return self.fusion_head(features)

    def forward(self, x):
        return self._tail(self.fused_embedding(x))

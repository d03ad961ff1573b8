"""Band-attention fusion heads: module names, constructor kwargs and state_dict keys of
/root/reference/main/models/multi_dino_attention.py (CrossAttentionBottleneckHead :1001-1062,
...Advanced :1064-1141, ...Pooled :484-599, ...Decoupled :336-481, get_fusion_head :602-690).

Eval-mode forward on GPU tensors runs the HIP/MFMA kernels (wv_band_attn_pool).  Training-mode
forward (dropout, autograd, ortho loss) is stock PyTorch on the GPU: training is outside the
accelerated path.  There is no CPU execution path.
"""
import ctypes
import logging

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

LOGGER = logging.getLogger("RETRIEVAL")


def _stacked(features_list):
    """S x [B, E] -> one [S, B, E] fp32 tensor.  Band features that already lie back to back in one allocation
    (slices of a preallocated [S, B, E] buffer the backbones wrote into) are used in place, without a copy."""
    f0 = features_list[0]
    if f0.dtype == torch.float32 and f0.dim() == 2 and f0.is_contiguous():
        base, step = f0.untyped_storage().data_ptr(), f0.numel() * 4
        if all(f.dtype == torch.float32 and f.shape == f0.shape and f.is_contiguous()
               and f.untyped_storage().data_ptr() == base and f.data_ptr() == f0.data_ptr() + i * step
               for i, f in enumerate(features_list)):
            return f0.as_strided((len(features_list),) + tuple(f0.shape), (f0.numel(), f0.shape[1], 1))
    return torch.stack([f.float() for f in features_list], dim=0).contiguous()


def band_attn_pool(features_list, q_eff, attn, norm1, norm2, mlp0, mlp2, out_proj, pool_mean=False,
                   workspace=None, qproj_cache=None, qproj_key=None):
    """HIP forward of the attention-pooling core.  features_list: S x [B, E] CUDA fp32.
    qproj_cache: a dict owned by the module; holds what is made from parameters alone -- the projected query tokens and,
    for the configurations with a one-launch front (wv_band_attn_prepared_bytes), the fragment-ordered copy of the
    weights -- so that it is rebuilt when a parameter changes, not in every call; qproj_key identifies the parameters
    q_eff was made from (storage pointers + version counters)."""
    lib = _lib.require_gpu()
    feats = _stacked(features_list)                                                   # [S, B, E]
    S, B, E = feats.shape
    q_src = q_eff
    q_eff = q_eff.detach().float().reshape(-1, E).contiguous()
    tensors = [q_eff, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias,
               norm1.weight, norm1.bias, mlp0.weight, mlp0.bias, mlp2.weight, mlp2.bias,
               out_proj.weight, out_proj.bias, norm2.weight, norm2.bias]
    keep = [t.detach().float().contiguous() for t in tensors]
    for t in keep:
        if not t.is_cuda:
            raise ValueError("band_attn_pool: module parameters must live on the GPU")
    p = _lib.HeadParams()
    p.embed_dim, p.num_heads, p.num_queries, p.num_tokens = E, attn.num_heads, q_eff.shape[0], S
    p.pool_mean = 1 if pool_mean else 0
    (p.q_eff, p.in_proj_w, p.in_proj_b, p.attn_out_w, p.attn_out_b, p.norm1_w, p.norm1_b, p.mlp0_w, p.mlp0_b,
     p.mlp2_w, p.mlp2_b, p.out_w, p.out_b, p.norm2_w, p.norm2_b) = [t.data_ptr() for t in keep]
    p.ln_eps = float(norm1.eps)
    p.q_proj = None
    out = torch.empty((B, E), dtype=torch.float32, device=feats.device)
    if B == 0:
        return out
    p.prepared = None
    if qproj_cache is not None:
        # the query tokens and the weights are parameters: key on their storage and version counters
        watched = (attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, mlp0.weight, mlp2.weight)
        key = (qproj_key if qproj_key is not None else (q_src.data_ptr(), q_src._version),
               tuple((t.data_ptr(), t._version) for t in watched), attn.num_heads, S, feats.device)
        if qproj_cache.get("key") != key:
            with torch.cuda.device(feats.device):
                nbytes = lib.wv_band_attn_prepared_bytes(ctypes.byref(p))
                if nbytes:      # this configuration has the one-launch front: projected queries + fragment-ordered weights
                    blob = torch.empty(nbytes, dtype=torch.uint8, device=feats.device)
                    _lib.check(lib.wv_band_attn_prepare(ctypes.byref(p), _lib.ptr(blob), _lib.stream_ptr()), "wv_band_attn_prepare")
                    entry = dict(key=key, blob=blob, qp=None)
                else:
                    qp = torch.empty_like(q_eff)
                    _lib.check(lib.wv_band_attn_qproj(ctypes.byref(p), _lib.ptr(qp), _lib.stream_ptr()), "wv_band_attn_qproj")
                    entry = dict(key=key, blob=None, qp=qp)
            qproj_cache.clear()
            qproj_cache.update(entry)
        if qproj_cache["blob"] is not None:
            p.prepared = qproj_cache["blob"].data_ptr()
        else:
            p.q_proj = qproj_cache["qp"].data_ptr()
    ws_bytes = lib.wv_band_attn_pool_workspace_bytes(ctypes.byref(p), B)
    if workspace is None or workspace.numel() < ws_bytes or workspace.device != feats.device:
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=feats.device)
    with torch.cuda.device(feats.device):
        rc = lib.wv_band_attn_pool(ctypes.byref(p), _lib.ptr(feats), B, _lib.ptr(out), _lib.ptr(workspace),
                                   ctypes.c_size_t(workspace.numel()), _lib.stream_ptr())
        _lib.check(rc, "wv_band_attn_pool")
    return out


def band_attn_pool_host(features_list, q_eff, attn, norm1, norm2, mlp0, mlp2, out_proj, pool_mean=False):
    """The same forward on HOST tensors through the library's host twin (wv_band_attn_pool_cpu, csrc/host_head.cpp): for a
    model that was moved to the CPU on purpose.  fp32, machine-independent summation order; agrees with the kernels to
    fp32 rounding."""
    lib = _lib.load()
    feats = torch.stack([f.detach().float() for f in features_list], dim=0).contiguous()        # [S, B, E]
    S, B, E = feats.shape
    q_eff = q_eff.detach().float().reshape(-1, E).contiguous()
    keep = [t.detach().float().contiguous() for t in
            (q_eff, attn.in_proj_weight, attn.in_proj_bias, attn.out_proj.weight, attn.out_proj.bias, norm1.weight, norm1.bias,
             mlp0.weight, mlp0.bias, mlp2.weight, mlp2.bias, out_proj.weight, out_proj.bias, norm2.weight, norm2.bias)]
    if any(t.is_cuda for t in keep) or feats.is_cuda:
        raise ValueError("band_attn_pool_host takes host tensors (features and parameters)")
    p = _lib.HeadParams()
    p.embed_dim, p.num_heads, p.num_queries, p.num_tokens = E, attn.num_heads, q_eff.shape[0], S
    p.pool_mean = 1 if pool_mean else 0
    (p.q_eff, p.in_proj_w, p.in_proj_b, p.attn_out_w, p.attn_out_b, p.norm1_w, p.norm1_b, p.mlp0_w, p.mlp0_b,
     p.mlp2_w, p.mlp2_b, p.out_w, p.out_b, p.norm2_w, p.norm2_b) = [t.data_ptr() for t in keep]
    p.ln_eps = float(norm1.eps)
    p.q_proj = None
    p.prepared = None
    out = torch.empty((B, E), dtype=torch.float32)
    _lib.check(lib.wv_band_attn_pool_cpu(ctypes.byref(p), _lib.ptr(feats), B, _lib.ptr(out)), "wv_band_attn_pool_cpu")
    return out


class CrossAttentionBottleneckHeadAdvanced(nn.Module):
    _pool = "concat"

    def __init__(self, input_dims, embed_dim=384, num_queries=4, num_heads=8, dropout=0.1,
                 sub_band_dropout_p=0.3, ortho_weight=0.1, margin=0.0, use_all_tokens=False):
        super().__init__()
        self.num_queries = num_queries
        self.sub_band_dropout_p = sub_band_dropout_p
        self.ortho_weight = ortho_weight
        self.margin = margin
        self.use_all_tokens = use_all_tokens
        self.projections = nn.ModuleList([
            nn.Linear(dim, embed_dim) if dim != embed_dim else nn.Identity() for dim in input_dims
        ])
        self.query_tokens = nn.Parameter(torch.randn(1, num_queries, embed_dim))
        nn.init.trunc_normal_(self.query_tokens, std=0.02)
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout, batch_first=True)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim, embed_dim * 4), nn.GELU(),
            nn.Linear(embed_dim * 4, embed_dim), nn.Dropout(dropout)
        )
        in_dim = embed_dim if self._pool == "mean" else num_queries * embed_dim
        self.out_proj = nn.Linear(in_dim, embed_dim)
        self.last_ortho_loss = 0.0
        self._ws = None
        self._zero_loss = None
        self._qproj_cache = {}

    # -- pieces shared by the four variants ------------------------------------------------
    def compute_ortho_loss(self):
        Q = self.query_tokens.squeeze(0)
        Q_norm = F.normalize(Q, p=2, dim=-1)
        gram = torch.matmul(Q_norm, Q_norm.T)
        identity = torch.eye(self.num_queries, device=Q.device)
        raw_error = torch.norm(gram - identity, p='fro')
        active_error = F.relu(raw_error - self.margin)
        return self.ortho_weight * (active_error ** 2)

    def effective_queries(self):
        return self.query_tokens

    def _query_key(self):
        """Identity of the parameters effective_queries() is made from (for the cached query projection)."""
        return tuple((t.data_ptr(), t._version) for t in (self.query_tokens,))

    def _ortho_after_attention(self, attn_weights, mask_ll, device):
        if self.training and self.ortho_weight > 0:
            return self.compute_ortho_loss()
        return torch.zeros((), device=device)

    def _readout(self, x, batch_size):
        return x.mean(dim=1) if self._pool == "mean" else x.view(batch_size, -1)

    def _hip_ok(self, kv_list):
        """Eval mode, CLS tokens only ([B, E] per band), nothing that needs autograd."""
        if self.training or self.use_all_tokens or len(kv_list) > 64:
            return False
        if any(t.dim() != 2 or not t.is_cuda for t in kv_list):
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in kv_list))

    def forward(self, features_list):
        batch_size = features_list[0].shape[0]
        device = features_list[0].device
        kv_list = [proj(f) for proj, f in zip(self.projections, features_list)]
        if not features_list[0].is_cuda and not self.training:
            # host tensors: never a silent fallback -- only after an explicit `head.host_twin = True` (a model meant to run
            # on the CPU) does the library's host twin take the eval-mode forward (same math, fp32)
            if not getattr(self, "host_twin", False):
                raise _lib.WvhashUnavailable("the eval-mode fusion head runs on the GPU; for a model that is meant to run on "
                                             "the host set `head.host_twin = True` (wv_band_attn_pool_cpu) -- there is no "
                                             "silent CPU fallback.  The training-mode forward is stock PyTorch and follows "
                                             "its tensors' device")
            if (self.use_all_tokens or len(kv_list) > 64 or any(t.dim() != 2 for t in kv_list)
                    or self.norm1.normalized_shape[0] % 8 or next(self.parameters()).is_cuda):
                raise _lib.WvhashUnavailable("the eval-mode fusion head on host tensors covers CLS-token inputs ([B, E] per "
                                             "band, E a multiple of 8, at most 64 bands) of a model that lives on the host")
            self.last_ortho_loss = torch.zeros(())
            with torch.no_grad():
                return band_attn_pool_host(kv_list, self.effective_queries(), self.attn, self.norm1, self.norm2,
                                           self.mlp[0], self.mlp[2], self.out_proj, self._pool == "mean")

        if self._hip_ok(kv_list):
            # torch.zeros launches a fill on the stream; torch.tensor(0.0, device=...) is a blocking host-to-device
            # copy that drains everything queued before it (here: the whole SWT kernel of the step)
            if self._zero_loss is None or self._zero_loss.device != device:
                self._zero_loss = torch.zeros((), device=device)
            self.last_ortho_loss = self._zero_loss
            with torch.no_grad():
                out = band_attn_pool(kv_list, self.effective_queries(), self.attn, self.norm1, self.norm2,
                                     self.mlp[0], self.mlp[2], self.out_proj, self._pool == "mean", self._ws,
                                     self._qproj_cache, self._query_key())
            return out

        # training / unsupported shapes: stock PyTorch on the GPU (outside the accelerated path)
        mask_ll = self.training and (torch.rand(1).item() < self.sub_band_dropout_p)
        if mask_ll:
            kv_list[0] = torch.zeros_like(kv_list[0])
        kv = torch.cat(kv_list, dim=1) if self.use_all_tokens else torch.stack(kv_list, dim=1)
        q = self.effective_queries().expand(batch_size, -1, -1)
        attn_output, attn_weights = self.attn(query=q, key=kv, value=kv)
        self.last_ortho_loss = self._ortho_after_attention(attn_weights, mask_ll, device)
        x = self.norm1(q + attn_output)
        x = x + self.mlp(x)
        x = self._readout(x, batch_size)
        x = self.out_proj(x)
        return self.norm2(x)


class CrossAttentionBottleneckHead(CrossAttentionBottleneckHeadAdvanced):
    """Ortho loss on the attention weights instead of the query Gram (:1047-1052)."""

    def __init__(self, input_dims, embed_dim=384, num_queries=4, num_heads=8, dropout=0.1,
                 sub_band_dropout_p=0.3, ortho_weight=0.1, use_all_tokens=False):
        super().__init__(input_dims, embed_dim, num_queries, num_heads, dropout, sub_band_dropout_p,
                         ortho_weight, 0.0, use_all_tokens)

    def _ortho_after_attention(self, attn_weights, mask_ll, device):
        if mask_ll or not self.training:
            return torch.tensor(0.0, device=device, requires_grad=True)
        M = attn_weights.mean(dim=0)
        identity = torch.eye(self.num_queries, device=device)
        return self.ortho_weight * (torch.norm(M @ M.t() - identity, p='fro') ** 2)


class CrossAttentionBottleneckHeadPooled(CrossAttentionBottleneckHeadAdvanced):
    def __init__(self, input_dims, embed_dim=384, num_queries=4, num_heads=8, dropout=0.1,
                 sub_band_dropout_p=0.3, ortho_weight=0.1, margin=0.0, use_all_tokens=False,
                 query_pool='mean'):
        if query_pool not in ('mean', 'concat'):
            raise ValueError(f"query_pool must be 'mean' or 'concat', got {query_pool!r}")
        self._pool = query_pool
        self.query_pool = query_pool
        super().__init__(input_dims, embed_dim, num_queries, num_heads, dropout, sub_band_dropout_p,
                         ortho_weight, margin, use_all_tokens)
        self.query_pool = query_pool


class CrossAttentionBottleneckHeadDecoupled(CrossAttentionBottleneckHeadAdvanced):
    def __init__(self, input_dims, embed_dim=384, num_queries=4, num_heads=8, dropout=0.1,
                 sub_band_dropout_p=0.3, ortho_weight=0.1, margin=0.0, use_all_tokens=False,
                 query_scale_init=4.0, normalize_queries=True, learn_query_scale=True):
        super().__init__(input_dims, embed_dim, num_queries, num_heads, dropout, sub_band_dropout_p,
                         ortho_weight, margin, use_all_tokens)
        self.normalize_queries = normalize_queries
        scale = torch.tensor(float(query_scale_init))
        if learn_query_scale:
            self.query_scale = nn.Parameter(scale)
        else:
            self.register_buffer('query_scale', scale)

    def effective_queries(self):
        q = self.query_tokens
        if self.normalize_queries:
            q = F.normalize(q, p=2, dim=-1)
        return q * self.query_scale

    def _query_key(self):
        return tuple((t.data_ptr(), t._version) for t in (self.query_tokens, self.query_scale)) + (self.normalize_queries,)


_HIP_TYPES = {
    'cross_attention_bottleneck': CrossAttentionBottleneckHead,
    'cross_attention_advanced': CrossAttentionBottleneckHeadAdvanced,
    'cross_attention_pooled': CrossAttentionBottleneckHeadPooled,
    'cross_attention_decoupled': CrossAttentionBottleneckHeadDecoupled,
}


def get_fusion_head(fusion_config, output_dims):
    """Same dispatch keys and defaults as the reference (:602-690)."""
    fusion_type = fusion_config.get('type', 'standard')
    embed_dim = fusion_config['output_dim']
    common = dict(
        num_queries=fusion_config.get('num_queries', 4),
        num_heads=fusion_config.get('num_heads', 8),
        dropout=fusion_config.get('dropout', 0.1),
        sub_band_dropout_p=fusion_config.get('sub_band_dropout_p', 0.3),
        ortho_weight=fusion_config.get('ortho_weight', 0.1),
    )
    if fusion_type in ('cross_attention_bottleneck', 'cross_attention_advanced'):
        return _HIP_TYPES[fusion_type](output_dims, embed_dim, **common)
    if fusion_type == 'cross_attention_pooled':
        return CrossAttentionBottleneckHeadPooled(
            output_dims, embed_dim, use_all_tokens=fusion_config.get('use_all_tokens', False),
            query_pool=fusion_config.get('query_pool', 'mean'), **common)
    if fusion_type == 'cross_attention_decoupled':
        return CrossAttentionBottleneckHeadDecoupled(
            output_dims, embed_dim, use_all_tokens=fusion_config.get('use_all_tokens', False),
            query_scale_init=fusion_config.get('query_scale_init', 4.0),
            normalize_queries=fusion_config.get('normalize_queries', True),
            learn_query_scale=fusion_config.get('learn_query_scale', True), **common)
    # every other type (standard, temperature, semantic, gated, temperature_gated, self_attention, cbam, eca; unknown
    # names fall back to standard like the reference): stock PyTorch modules with the reference's state_dict keys
    from .fusion_extra import build_extra_head
    LOGGER.info("fusion type '%s' runs as stock PyTorch (the HIP head covers the cross-attention family)", fusion_type)
    return build_extra_head(fusion_type, fusion_config, output_dims)

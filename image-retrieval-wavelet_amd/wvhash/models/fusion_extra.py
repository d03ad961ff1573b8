"""The reference's other band-fusion heads (everything `get_fusion_head` offers besides the cross-attention bottleneck family):
/root/reference/main/models/multi_dino_attention.py:156-334 -- `standard`, `temperature`, `semantic`, `gated`,
`temperature_gated`, `self_attention`, `cbam`, `eca`.

They are outside the accelerated path (the headline configurations use `cross_attention_*`, fusion.py): stock PyTorch on
whatever device the parameters live on, forward and backward.  What this module guarantees is the drop-in contract: class names,
constructor arguments, and state_dict keys of the reference, so its configs and checkpoints select and load them unchanged
(tests/test_fusion_extra.py loads this module's state_dict into the reference's classes, strictly, and compares outputs).

One skeleton serves all token heads: per-band projections -> a mixing rule that turns the S band vectors into one vector per
sample -> LayerNorm, GELU MLP with residual, LayerNorm.  The subclasses differ in the mixing rule only.
"""
import torch
import torch.nn as nn


def _band_projections(input_dims, embed_dim):
    return nn.ModuleList([nn.Identity() if d == embed_dim else nn.Linear(d, embed_dim) for d in input_dims])


def _ffn(embed_dim, dropout):
    return nn.Sequential(nn.Linear(embed_dim, 4 * embed_dim), nn.GELU(), nn.Linear(4 * embed_dim, embed_dim), nn.Dropout(dropout))


class _TokenHead(nn.Module):
    """projections / norm1 / norm2 / mlp, and the tail every head shares: y = LN2(z + MLP(z)), z = LN1(mixed)."""

    def __init__(self, input_dims, embed_dim, dropout):
        super().__init__()
        self.projections = _band_projections(input_dims, embed_dim)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.mlp = _ffn(embed_dim, dropout)

    def _project(self, features_list):
        return [p(f) for p, f in zip(self.projections, features_list)]

    def _tail(self, mixed):
        z = self.norm1(mixed)
        return self.norm2(z + self.mlp(z))

    def mix(self, bands):                       # list of S tensors [B, E] (or [B, T, E]) -> [B, E] (or [B, 1, E])
        raise NotImplementedError

    def forward(self, features_list):
        out = self._tail(self.mix(self._project(features_list)))
        return out.squeeze(1) if out.dim() == 3 else out


class _QueryTokenHead(_TokenHead):
    """One learned query token attends over the band tokens (nn.MultiheadAttention, batch_first)."""
    residual_query = False                      # LN1(q + attn) instead of LN1(attn)
    query_divisor = None                        # temperature: the query is divided by it before attention

    def __init__(self, input_dims, embed_dim=384, num_heads=8, dropout=0.1, use_all_tokens=False):
        super().__init__(input_dims, embed_dim, dropout)
        self.use_all_tokens = use_all_tokens
        self.query_token = nn.Parameter(torch.randn(1, 1, embed_dim))
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout, batch_first=True)
        nn.init.trunc_normal_(self.query_token, std=0.02)

    def _keys(self, bands):
        return torch.cat(bands, dim=1) if self.use_all_tokens else torch.stack(bands, dim=1)

    def mix(self, bands):
        kv = self._keys(bands)
        q = self.query_token.expand(kv.shape[0], -1, -1)
        asked = q / self.query_divisor if self.query_divisor is not None else q
        ctx, _ = self.attn(query=asked, key=kv, value=kv)
        return q + ctx if self.residual_query else ctx


class StandardFusionHead(_QueryTokenHead):
    """`standard` (and the fallback of get_fusion_head): multi_dino_attention.py:179-201."""


class AttentionFusionHead(_QueryTokenHead):
    """`self_attention`: as standard with the query token added back before LayerNorm 1 (:293-334)."""
    residual_query = True


class TemperatureFusionHead(_QueryTokenHead):
    """`temperature`: the query token is divided by a temperature, sharpening the band softmax (:204-225)."""

    def __init__(self, input_dims, embed_dim=384, num_heads=4, dropout=0.1, temperature=0.1):
        super().__init__(input_dims, embed_dim, num_heads, dropout)
        self.temperature = temperature
        self.query_divisor = temperature


class SemanticFusionHead(_TokenHead):
    """`semantic`: the first band (LL) asks, all bands answer (:227-243).  No learned query token."""

    def __init__(self, input_dims, embed_dim=512, num_heads=4, dropout=0.1):
        super().__init__(input_dims, embed_dim, dropout)
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout, batch_first=True)

    def mix(self, bands):
        kv = torch.stack(bands, dim=1)
        ctx, _ = self.attn(query=kv[:, :1], key=kv, value=kv)
        return ctx


class GatedFusionHead(_TokenHead):
    """`gated`: every band is weighted by a scalar gate computed from itself, the weighted bands are summed (:245-262)."""
    gate_temperature = None                     # None: the sigmoid is part of gate_network (the reference's layout)

    def __init__(self, input_dims, embed_dim=512, dropout=0.1):
        super().__init__(input_dims, embed_dim, dropout)
        layers = [nn.Linear(embed_dim, embed_dim // 2), nn.ReLU(), nn.Linear(embed_dim // 2, 1)]
        if self.gate_temperature is None:
            layers.append(nn.Sigmoid())
        self.gate_network = nn.Sequential(*layers)

    def _gate(self, band):
        g = self.gate_network(band)
        return g if self.gate_temperature is None else torch.sigmoid(g / self.gate_temperature)

    def mix(self, bands):
        total = bands[0] * self._gate(bands[0])
        for band in bands[1:]:
            total = total + band * self._gate(band)
        return total


class TemperatureGatedFusionHead(GatedFusionHead):
    """`temperature_gated`: the gate logit is divided by a temperature before the sigmoid (:264-291)."""
    gate_temperature = 0.1

    def __init__(self, input_dims, embed_dim=512, dropout=0.1, temperature=0.1):
        self.gate_temperature = temperature     # before super().__init__: decides whether gate_network ends in a Sigmoid
        super().__init__(input_dims, embed_dim, dropout)
        self.temperature = temperature


# ------------------------------------------------------------------------------------------ branch gates (cbam / eca)
class ChannelGate(nn.Module):
    """Squeeze-and-excite over the BRANCH axis: per-branch statistics over the feature axis (average, maximum) go through
    one small MLP, their sum through a sigmoid, and the branches are averaged with those weights (:33-91, the 1-D use the
    fusion module makes of it).  state_dict: mlp.1.*, mlp.3.* (index 0 is the parameter-free flatten)."""

    def __init__(self, gate_channels, reduction_ratio=1, pool_types=("avg", "max")):
        super().__init__()
        self.gate_channels = gate_channels
        self.pool_types = tuple(pool_types)
        hidden = gate_channels // reduction_ratio
        self.mlp = nn.Sequential(nn.Flatten(), nn.Linear(gate_channels, hidden), nn.ReLU(), nn.Linear(hidden, gate_channels))

    def alphas(self, x):                        # x [B, branches, D] -> [B, branches, 1]
        stats = {"avg": lambda t: t.mean(dim=2), "max": lambda t: t.amax(dim=2),
                 "lse": lambda t: torch.logsumexp(t, dim=2), "lp": lambda t: t.pow(2).sum(dim=2).sqrt()}
        logits = sum(self.mlp(stats[kind](x)) for kind in self.pool_types)
        return torch.sigmoid(logits).unsqueeze(-1)

    def forward(self, x):
        return (x * self.alphas(x)).sum(dim=1) / self.gate_channels


class CBAM(nn.Module):
    """Branch gate of the `cbam` fusion (no_spatial: the channel gate alone, :116-134)."""

    def __init__(self, gate_channels=4, reduction_ratio=1, pool_types=("avg", "max"), no_spatial=True):
        super().__init__()
        if not no_spatial:
            raise NotImplementedError("the fusion module uses CBAM without its spatial gate")
        self.ChannelGate = ChannelGate(gate_channels, reduction_ratio, pool_types)
        self.no_spatial = True

    def alphas(self, x):
        return self.ChannelGate.alphas(x)

    def forward(self, x):
        return self.ChannelGate(x)


class Eca1D_layer(nn.Module):
    """Efficient channel attention over the branch axis: branch means -> a 3-tap convolution ALONG the branches -> sigmoid
    weights (:136-154).  state_dict: conv.weight."""

    def __init__(self, channel, k_size=3):
        super().__init__()
        self.chan = channel
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)

    def alphas(self, x):                        # x [B, branches, D] -> [B, branches, 1]
        means = x.mean(dim=2, keepdim=True)     # [B, branches, 1]
        return torch.sigmoid(self.conv(means.transpose(1, 2)).transpose(1, 2))

    def forward(self, x):
        return (x * self.alphas(x)).sum(dim=1) / self.chan


class AdvancedFusionModule(nn.Module):
    """`cbam` / `eca`: gate-weighted average of the branch embeddings, then Linear -> BatchNorm1d -> ReLU -> Dropout (:156-176)."""

    def __init__(self, fusion_type="cbam", num_branches=4, reduction_ratio=1, input_dim=384, hidden_dim=384):
        super().__init__()
        if fusion_type == "cbam":
            self.gate = CBAM(gate_channels=num_branches, reduction_ratio=reduction_ratio, pool_types=("avg", "max"), no_spatial=True)
        elif fusion_type == "eca":
            self.gate = Eca1D_layer(channel=num_branches, k_size=3)
        else:
            raise ValueError(f"AdvancedFusionModule: unknown fusion_type {fusion_type!r}")
        self.fcn = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.ReLU(inplace=True), nn.Dropout(p=0.1))

    def forward(self, embeddings_list):
        return self.fcn(self.gate(torch.stack(embeddings_list, dim=1)))


def build_extra_head(fusion_type, fusion_config, output_dims):
    """The non-cross-attention branches of get_fusion_head (multi_dino_attention.py:602-690), same defaults."""
    embed_dim = fusion_config["output_dim"]
    heads = fusion_config.get("num_heads", 8)
    dropout = fusion_config.get("dropout", 0.1)
    if fusion_type == "temperature":
        return TemperatureFusionHead(output_dims, embed_dim, heads, dropout, temperature=fusion_config.get("temperature", 0.1))
    if fusion_type == "semantic":
        return SemanticFusionHead(output_dims, embed_dim, heads, dropout)
    if fusion_type == "gated":
        return GatedFusionHead(output_dims, embed_dim, dropout)
    if fusion_type == "temperature_gated":
        return TemperatureGatedFusionHead(output_dims, embed_dim, dropout, temperature=fusion_config.get("temperature", 0.1))
    if fusion_type == "self_attention":
        return AttentionFusionHead(output_dims, embed_dim, heads, dropout)
    if fusion_type in ("cbam", "eca"):
        return AdvancedFusionModule(fusion_type=fusion_type, num_branches=len(output_dims), input_dim=output_dims[0], hidden_dim=embed_dim)
    return StandardFusionHead(output_dims, embed_dim, heads, dropout)     # 'standard' and anything unknown, like the reference

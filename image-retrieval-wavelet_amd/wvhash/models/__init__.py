from .fusion import (CrossAttentionBottleneckHead, CrossAttentionBottleneckHeadAdvanced,
                     CrossAttentionBottleneckHeadPooled, CrossAttentionBottleneckHeadDecoupled,
                     get_fusion_head, band_attn_pool)
from .fusion_extra import (AdvancedFusionModule, AttentionFusionHead, GatedFusionHead, SemanticFusionHead, StandardFusionHead,
                           TemperatureFusionHead, TemperatureGatedFusionHead)
from .hashing import SharedDinoHashing, MultiDinoHashing, hash_tail, load_dinov2

__all__ = ["CrossAttentionBottleneckHead", "CrossAttentionBottleneckHeadAdvanced",
           "CrossAttentionBottleneckHeadPooled", "CrossAttentionBottleneckHeadDecoupled", "get_fusion_head",
           "band_attn_pool", "SharedDinoHashing", "MultiDinoHashing", "hash_tail", "load_dinov2", "StandardFusionHead",
           "TemperatureFusionHead", "SemanticFusionHead", "GatedFusionHead", "TemperatureGatedFusionHead",
           "AttentionFusionHead", "AdvancedFusionModule"]

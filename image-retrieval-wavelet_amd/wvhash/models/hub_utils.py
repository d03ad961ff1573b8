"""DINOv2 loader with the call signature of /root/reference/main/models/hub_utils.py:35-59.

The backbone is not part of the accelerated path; it is fetched with torch.hub exactly like the
reference does.  Offline (no hub cache, no network) this raises -- pass ``backbone=`` to the model
classes instead (bench / tests use wvhash.models.vit.TinyViT or a random-init ViT-S/14).
"""
import torch


def load_dinov2(name, retries=1, **kwargs):
    last = None
    for _ in range(max(1, retries)):
        try:
            return torch.hub.load("facebookresearch/dinov2:main", name, skip_validation=True, **kwargs)
        except Exception as e:  # network / cache miss
            last = e
    raise RuntimeError(
        f"load_dinov2('{name}') failed ({type(last).__name__}: {last}). torch.hub needs network or a "
        "populated hub cache; construct the model with backbone=<nn.Module with .embed_dim> instead.")

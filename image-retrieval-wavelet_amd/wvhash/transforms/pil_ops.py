"""Host-side image sizing ops that sit in front of the wavelet transform in the reference's YAML pipelines
(config/transform/basic_swt.yaml: Resize -> CenterCrop -> SWTTransform), and the plugin lookup that builds the
pipeline (Getter.get_transform, /root/reference/main/getter.py:25-35).

The reference takes Resize / CenterCrop / Compose from torchvision; torchvision is an optional dependency here
(absent -> these PIL equivalents are used: same output size rule, PIL bilinear filter like torchvision applies to
PIL images).  They decode and size images on the host; the arithmetic of the path (SWT) stays on the GPU.
"""
from PIL import Image


class Resize(object):
    """int size: the SHORTER side becomes `size`, aspect kept (long side = int(size * long / short));
    (h, w): exact size.  torchvision.transforms.Resize semantics for PIL input, bilinear."""

    def __init__(self, size, interpolation=Image.BILINEAR, **kwargs):
        self.size = size
        self.interpolation = interpolation

    def __call__(self, img):
        if isinstance(self.size, int) or (hasattr(self.size, "__len__") and len(self.size) == 1):
            s = self.size if isinstance(self.size, int) else self.size[0]
            w, h = img.size
            short, long = (w, h) if w <= h else (h, w)
            if short == s:
                return img
            new_short, new_long = s, int(s * long / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            return img.resize((nw, nh), self.interpolation)
        h, w = self.size
        return img.resize((int(w), int(h)), self.interpolation)

    def __repr__(self):
        return f"Resize(size={self.size})"


class CenterCrop(object):
    """Crop (or zero-pad, when the image is smaller) to `size` around the centre, torchvision rounding."""

    def __init__(self, size, **kwargs):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        th, tw = self.size
        w, h = img.size
        if w < tw or h < th:  # torchvision pads with zeros first
            canvas = Image.new(img.mode, (max(w, tw), max(h, th)))
            canvas.paste(img, ((max(w, tw) - w) // 2, (max(h, th) - h) // 2))
            img, (w, h) = canvas, canvas.size
        top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        return img.crop((left, top, left + tw, top + th))

    def __repr__(self):
        return f"CenterCrop(size={self.size})"


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img

    def __repr__(self):
        return "Compose(\n" + "\n".join(f"    {t}" for t in self.transforms) + "\n)"


def build_transform(config, defer=False, device=None):
    """``{name: kwargs}`` mapping (a YAML node) -> Compose, resolved like Getter.get_transform: wavelet plugins of
    this package first, then torchvision.transforms when importable, else the PIL equivalents above.
    defer / device are forwarded to the wavelet plugins (see custom_transforms.py)."""
    from . import custom_transforms as ct
    try:
        import torchvision.transforms as tvt
    except ImportError:
        tvt = None
    local = {"Resize": Resize, "CenterCrop": CenterCrop}
    steps = []
    for name, kwargs in dict(config).items():
        kwargs = dict(kwargs or {})
        if hasattr(ct, name) and name.endswith("Transform"):
            steps.append(getattr(ct, name)(defer=defer, device=device, **kwargs))
        elif tvt is not None and hasattr(tvt, name):
            steps.append(getattr(tvt, name)(**kwargs))
        elif name in local:
            steps.append(local[name](**kwargs))
        else:
            raise AttributeError(f"transform '{name}' is neither a wvhash plugin nor available without torchvision")
    return Compose(steps)

"""Drop-in wavelet transform plugins: same class names, kwargs, output layout and repr as
/root/reference/main/transforms/custom_transforms.py:126-205.

Where ``SWTTransform(level, wavelet)(pil_img)`` computes (the result is always ``FloatTensor[3, 4, H', W']``):
* inside a DataLoader worker process (``torch.utils.data.get_worker_info()`` is set) -- the way the reference's
  unchanged YAML + ``DataLoader(num_workers > 0)`` calls it (flikr_coco.py:59-60) -- or with ``device="cpu"``:
  the library's host twin ``wv_swt2d_forward_cpu`` (single-threaded native code, no HIP call, fork-safe); a forked
  worker cannot use the parent's GPU context;
* in a process that owns the device: the HIP kernel (one image per call);
* no GPU and no ``device="cpu"``: ``WvhashUnavailable`` -- the transform never silently changes device.
The fast way to run the path is batched: construct with ``defer=True`` (``build_transform(cfg, defer=True)``); the
workers then only size the image and return the raw ``uint8 [3, H', W']`` tensor (16x fewer bytes over PCIe than
the expanded fp32 sub-bands), and the evaluation engine / the model run ``transform.apply_batch`` on the GPU after
collate.  A deferred batch carries no wavelet name: ``engine.evaluate`` reads it from ``dataset.transform`` and binds
the transform to the model (``model.bind_transform``); a model that receives a raw 4-D batch without a bound
transform raises instead of guessing.
"""
import numpy as np
import torch
from PIL import Image

from . import functional as F


class BaseWaveletTransform(object):
    """Shared wavelet pipeline: resize to fit the level, transform each RGB channel, stack."""

    def __init__(self, level=1, wavelet='haar', defer=False, device=None):
        self.level = level
        self.wavelet = wavelet
        self.defer = defer
        self.device = device

    def fix_size(self, image):
        w, h = image.size
        factor = 2 ** self.level
        new_w = int(np.ceil(w / factor) * factor)
        new_h = int(np.ceil(h / factor) * factor)
        if new_w != w or new_h != h:
            image = image.resize((new_w, new_h), resample=Image.BICUBIC)
        return image

    def _to_u8_hwc(self, img):
        if isinstance(img, Image.Image):
            img = self.fix_size(img)
            arr = np.array(img)
        else:
            arr = np.asarray(img)
        if arr.ndim != 3 or arr.shape[2] < 3:
            raise ValueError(f"expected an RGB image (H, W, 3), got array of shape {arr.shape}")
        return np.ascontiguousarray(arr[:, :, :3])

    def _device(self):
        return torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())

    def apply_batch(self, batch, channels_last=False, **kwargs):
        """Batched path.  batch: uint8/float32 [B,3,H,W] (or [B,H,W,3]); GPU tensors take the HIP kernel, host
        tensors the host twin."""
        raise NotImplementedError

    def _use_host(self):
        """True when __call__ must not touch the GPU: explicit device='cpu', or a DataLoader worker process."""
        if self.device is not None and torch.device(self.device).type == "cpu":
            return True
        from torch.utils.data import get_worker_info
        return get_worker_info() is not None

    def __call__(self, img):
        arr = self._to_u8_hwc(img)
        if self.defer:
            if arr.dtype != np.uint8:
                raise TypeError("deferred transforms expect 8-bit images")
            return torch.from_numpy(arr).permute(2, 0, 1).contiguous()
        if arr.dtype == np.uint8:
            x = torch.from_numpy(arr)
        else:  # the reference computes astype(float32) / 255 for any dtype
            x = torch.from_numpy(arr.astype(np.float32) / 255.0)
        x = x.unsqueeze(0)
        if self._use_host():
            return self.apply_batch(x, channels_last=True)[0]
        from .. import _lib
        try:
            _lib.require_gpu()
        except _lib.WvhashUnavailable as e:
            raise _lib.WvhashUnavailable(f"{e}; pass device='cpu' to run the library's host twin in this process "
                                         "(DataLoader workers take it automatically)") from None
        return self.apply_batch(x.to(self._device(), non_blocking=True), channels_last=True)[0].cpu().float()

    def deferred_spec(self):
        """What a deferred batch needs to be expanded later: (kind, wavelet, level, copies)."""
        return (type(self).__name__, self.wavelet, self.level, getattr(self, "copies", 4))


def find_wavelet_transform(obj):
    """The wavelet plugin inside a transform pipeline (a plugin itself, a Compose-like object with a
    ``transforms`` list, a dataset with a ``transform`` attribute, or a wrapper such as torch.utils.data.Subset holding
    one in ``dataset``); None when there is none."""
    if isinstance(obj, BaseWaveletTransform):
        return obj
    for attr in ("transform", "transforms", "dataset"):
        inner = getattr(obj, attr, None)
        if inner is None:
            continue
        if isinstance(inner, (list, tuple)):
            for t in inner:
                found = find_wavelet_transform(t)
                if found is not None:
                    return found
        else:
            found = find_wavelet_transform(inner)
            if found is not None:
                return found
    return None


class SWTTransform(BaseWaveletTransform):
    """Stationary wavelet transform (size preserved: H, W)."""

    def apply_batch(self, batch, channels_last=False, **kwargs):
        if not batch.is_cuda:
            return F.swt2d_host(batch, self.wavelet, self.level, channels_last=channels_last)
        return F.swt2d(batch, self.wavelet, self.level, channels_last=channels_last, **kwargs)

    def __repr__(self):
        return f"SWTTransform(shape='C,S,H,W', wavelet={self.wavelet}, level={self.level})"


class RawStackTransform(BaseWaveletTransform):
    """Parameter-matched control: every 'subband' is an identical copy of the raw channel."""

    def __init__(self, level=1, wavelet='haar', copies=4, defer=False, device=None):
        super().__init__(level=level, wavelet=wavelet, defer=defer, device=device)
        self.copies = copies

    def apply_batch(self, batch, channels_last=False, **kwargs):
        if not batch.is_cuda:
            return F.rawstack_host(batch, self.copies, channels_last=channels_last)
        return F.rawstack(batch, self.copies, channels_last=channels_last)

    def __repr__(self):
        return f"RawStackTransform(shape='C,{self.copies},H,W', copies={self.copies})"


class DWTTransform(BaseWaveletTransform):
    """Discrete multi-level wavelet transform (size divided by 2^level): the coarsest-level bands of
    ``pywt.wavedec2`` with PyWavelets' default symmetric extension (a plain, untuned kernel: this
    transform is outside the hot path)."""

    def __init__(self, level=1, wavelet='haar', defer=False, device=None):
        super().__init__(level=level, wavelet=wavelet, defer=defer, device=device)

    def apply_batch(self, batch, channels_last=False, **kwargs):
        if not batch.is_cuda:
            return F.dwt2d_host(batch, self.wavelet, self.level, channels_last=channels_last)
        return F.dwt2d(batch, self.wavelet, self.level, channels_last=channels_last)

    def __repr__(self):
        factor = 2 ** self.level
        return f"DWTTransform(shape='C,S,H/{factor},W/{factor}', wavelet={self.wavelet}, level={self.level})"

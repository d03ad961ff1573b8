"""Drop-in wavelet transform plugins: same class names, kwargs, output layout and repr as
/root/reference/main/transforms/custom_transforms.py:126-205, computed by the HIP kernels.

Two ways to use them
* per image, exactly like the reference: ``SWTTransform(level, wavelet)(pil_img)`` ->
  ``FloatTensor[3, 4, H', W']`` (computed on the GPU; the call must run in a process that owns
  the device, i.e. ``num_workers=0``);
* batched (how the path is meant to run): construct with ``defer=True``; inside DataLoader
  workers the transform then only sizes the image and returns the raw ``uint8 [3, H', W']``
  tensor (no GPU touched in the worker), and ``transform.apply_batch(batch_u8_on_gpu)`` -- called
  by the evaluation engine / the model wrapper after collate -- runs the one batched kernel.
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

    def apply_batch(self, batch, channels_last=False):
        """Batched device path.  batch: uint8/float32 [B,3,H,W] (or [B,H,W,3]) on the GPU."""
        raise NotImplementedError

    def __call__(self, img):
        arr = self._to_u8_hwc(img)
        if self.defer:
            if arr.dtype != np.uint8:
                raise TypeError("deferred transforms expect 8-bit images")
            return torch.from_numpy(arr).permute(2, 0, 1).contiguous()
        from .. import _lib
        _lib.require_gpu()
        if arr.dtype == np.uint8:
            x = torch.from_numpy(arr)
        else:  # the reference computes astype(float32) / 255 for any dtype
            x = torch.from_numpy(arr.astype(np.float32) / 255.0)
        x = x.unsqueeze(0).to(self._device(), non_blocking=True)
        return self.apply_batch(x, channels_last=True)[0].cpu().float()


class SWTTransform(BaseWaveletTransform):
    """Stationary wavelet transform (size preserved: H, W)."""

    def apply_batch(self, batch, channels_last=False):
        return F.swt2d(batch, self.wavelet, self.level, channels_last=channels_last)

    def __repr__(self):
        return f"SWTTransform(shape='C,S,H,W', wavelet={self.wavelet}, level={self.level})"


class RawStackTransform(BaseWaveletTransform):
    """Parameter-matched control: every 'subband' is an identical copy of the raw channel."""

    def __init__(self, level=1, wavelet='haar', copies=4, defer=False, device=None):
        super().__init__(level=level, wavelet=wavelet, defer=defer, device=device)
        self.copies = copies

    def apply_batch(self, batch, channels_last=False):
        return F.rawstack(batch, self.copies, channels_last=channels_last)

    def __repr__(self):
        return f"RawStackTransform(shape='C,{self.copies},H,W', copies={self.copies})"


class DWTTransform(BaseWaveletTransform):
    """Discrete multi-level wavelet transform (size divided by 2^level): the coarsest-level bands of
    ``pywt.wavedec2`` with PyWavelets' default symmetric extension (a plain, untuned kernel: this
    transform is outside the hot path)."""

    def __init__(self, level=1, wavelet='haar', defer=False, device=None):
        super().__init__(level=level, wavelet=wavelet, defer=defer, device=device)

    def apply_batch(self, batch, channels_last=False):
        return F.dwt2d(batch, self.wavelet, self.level, channels_last=channels_last)

    def __repr__(self):
        factor = 2 ** self.level
        return f"DWTTransform(shape='C,S,H/{factor},W/{factor}', wavelet={self.wavelet}, level={self.level})"

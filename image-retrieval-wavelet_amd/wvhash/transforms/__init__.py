from .custom_transforms import (SWTTransform, DWTTransform, RawStackTransform, BaseWaveletTransform,
                                find_wavelet_transform)
from .functional import swt2d, dwt2d, rawstack, swt2d_host, dwt2d_host, rawstack_host, swt2d_place_output
from .wavelets import get_filters, wavelist
from .pil_ops import Resize, CenterCrop, Compose, build_transform
from .lifting import CustomTransform, ResizeSubBands, HaarLifting, Cdf97Lifting

__all__ = ["SWTTransform", "DWTTransform", "RawStackTransform", "BaseWaveletTransform", "find_wavelet_transform", "swt2d", "swt2d_place_output", "swt2d_host", "dwt2d_host", "rawstack_host",
           "dwt2d", "rawstack", "get_filters", "wavelist", "Resize", "CenterCrop", "Compose", "build_transform", "CustomTransform", "ResizeSubBands", "HaarLifting",
           "Cdf97Lifting"]

#!/usr/bin/env python3
"""Phase breakdown of the fused head front (diagnostic build: tools/build_variant.sh hfstamps head_front.hip -DWV_HF_STAMPS,
run with WVHASH_LIB=tools/_variants/hfstamps.so).  Shader-clock cycles of wave 0 per phase, averaged over the workgroups."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import _lib, synth  # noqa: E402
from wvhash.models import get_fusion_head  # noqa: E402

NAMES = ["feats -> LDS", "scores product", "softmax", "V product", "ctx mix", "out product", "LayerNorm 1", "mlp.0 products (4)",
         "GELU + barrier (4)", "mlp.2 products (4)", "x2 store"]


def main():
    os.environ["WV_HEAD_FRONT"] = "1"
    lib = _lib.load()
    fn = lib.wv_debug_hf_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    buf = (ctypes.c_ulonglong * 16)()
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4, "num_heads": 8}, [384] * 4)
    head.load_state_dict(synth.head_state(384, 4, "concat", seed=0))
    head = head.cuda().eval()
    for B in (8, 2048, 4096):
        feats = [f.cuda() for f in synth.band_features(B, 384, seed=B)]
        with torch.no_grad():
            for _ in range(3):
                head(feats)
            torch.cuda.synchronize()
            fn(buf)
            head(feats)
            torch.cuda.synchronize()
            fn(buf)
        wgs = (B + 7) // 8
        tot = sum(buf[:11])
        print(f"B={B} ({wgs} workgroups): {tot / wgs:.0f} cycles per workgroup (s_memtime ticks)")
        for i, n in enumerate(NAMES):
            print(f"    {n:24s} {buf[i] / wgs:9.0f}  {100.0 * buf[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()

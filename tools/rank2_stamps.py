#!/usr/bin/env python3
"""Phase breakdown of the windowed ranking kernel (diagnostic build: tools/build_variant.sh r2stamps rank2.hip -DWV_RANK2_STAMPS,
run with WVHASH_LIB=tools/_variants/r2stamps.so).  Cycles of wave 0 of every query group, summed, per phase."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import _lib, synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402

NAMES = ["phase 0 distances", "zero + count", "totals + scans", "placement", "loop tail / cum", "copy-out + dist row",
         "relevance bitmap (fused AP)", "AP walk (fused AP)"]


def main():
    lib = _lib.load()
    fn = lib.wv_debug_rank2_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    buf = (ctypes.c_ulonglong * 8)()
    for name, Q, N, nbits, k, cum in [("c1", 2048, 25000, 64, 5000, False), ("shard 1/8", 16384, 3125, 64, 3125, True),
                                      ("shard 1/8 no cum", 16384, 3125, 64, 3125, False)]:
        ql, rl = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
        q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
        qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
        for variant, qb in (("256", "1"), ("256", "8"), ("64", "8")):
            os.environ["WV_TOPK_V2"] = variant
            os.environ["WV_TOPK_QB"] = qb
            try:
                H.hamming_topk(qp, prep, nbits, k, want_dist=not cum, want_cum=cum)
            except Exception:  # noqa: BLE001
                continue
            torch.cuda.synchronize()
            fn(buf)
            H.hamming_topk(qp, prep, nbits, k, want_dist=not cum, want_cum=cum)
            torch.cuda.synchronize()
            fn(buf)
            tot = sum(buf[:8])
            print(f"{name} variant {variant} qb {qb}: total {tot / Q:.0f} cycles/query: " +
                  ", ".join(f"{n} {buf[i] / Q:.0f}" for i, n in enumerate(NAMES)), flush=True)


def fused():
    """the same phases with the list evaluated in LDS (wv_hamming_map_at_k)"""
    lib = _lib.load()
    fn = lib.wv_debug_rank2_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    buf = (ctypes.c_ulonglong * 8)()
    Q, N, nbits, k = 2048, 25000, 64, 5000
    ql, rl = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    qlp, labels = H.pack_labels(ql.cuda()), H.PreparedLabels(H.pack_labels(rl.cuda()))
    for _ in range(2):
        H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)
        torch.cuda.synchronize()
        fn(buf)
    tot = sum(buf[:8])
    print(f"c1 fused AP: total {tot / Q:.0f} cycles/query: " + ", ".join(f"{n} {buf[i] / Q:.0f}" for i, n in enumerate(NAMES)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fused":
        fused()
        sys.exit(0)
    main()

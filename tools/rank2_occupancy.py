#!/usr/bin/env python3
"""Is a phase of the windowed ranking kernel bound by a per-CU resource (its time grows with the workgroups resident on the
CU) or by latency (it does not)?  Phase stamps (diagnostic build: tools/build_variant.sh r2stamps rank2.hip -DWV_RANK2_STAMPS;
run with WVHASH_LIB=tools/_variants/r2stamps.so) at 1 ... 5 resident workgroups per CU (WV_R2_PAD_LDS pads the dynamic LDS)
and one wave of queries (Q = 256 x residency), plus the launch time."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import _lib, synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402

NAMES = ["dist", "zero+count", "totals+scans", "placement", "tail/cum", "copy-out", "bitmap", "AP walk"]


def main():
    lib = _lib.load()
    fn = lib.wv_debug_rank2_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    buf = (ctypes.c_ulonglong * 8)()
    N, nbits, k = 25000, 64, 5000
    rl = synth.multi_hot_labels(N, 38, 0.10, 2)
    r = synth.structured_codes(rl, nbits, 3, 5)
    prep = H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    lds_q = 32 * 1024                                           # ~ what a query's workgroup takes at k = 5000
    for resident in (1, 2, 3, 4, 5):
        pad = max(0, 160 * 1024 // resident - lds_q - 1024) if resident < 5 else 0
        os.environ["WV_R2_PAD_LDS"] = str(pad)
        for rounds in (1, 2):
            Q = 256 * resident * rounds
            ql = synth.multi_hot_labels(Q, 38, 0.10, 1)
            qp = H.pack_codes(synth.structured_codes(ql, nbits, 3, 4).cuda())
            for _ in range(3):
                H.hamming_topk(qp, prep, nbits, k, want_dist=False)
            torch.cuda.synchronize()
            fn(buf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                H.hamming_topk(qp, prep, nbits, k, want_dist=False)
            e1.record()
            torch.cuda.synchronize()
            fn(buf)
            print(f"resident {resident} x {rounds} round(s), Q={Q:5d}: {e0.elapsed_time(e1) * 100:7.1f} us/launch; cycles/query: "
                  + ", ".join(f"{n} {buf[i] / Q / 10:.0f}" for i, n in enumerate(NAMES) if buf[i]), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Phases of k_row_topk by truncation: variant build with -DWV_TK_STOPS (tools/build_variant.sh tkstops knn_float.hip
-DWV_TK_STOPS), run under rocprofv3 --kernel-trace with WVHASH_LIB=tools/_variants/tkstops.so; the kernel ends after phase
WV_TK_STOP (1 sample + bound, 2 histogram pass, 3 scan, 4 placement, 5 in-bin ranks, 0 whole kernel).  Prints the order of
the settings; tools/rocprof_kernels.py --seq on the database gives the k_row_topk durations in the same order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import _lib  # noqa: E402
from wvhash.engine.get_knn import knn_float  # noqa: E402

REPS = 5
if __name__ == "__main__":
    torch.cuda.set_device(0)
    Q, N, D, k = 2048, 25000, int(sys.argv[2]) if len(sys.argv) > 2 else 384, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    g = torch.Generator().manual_seed(7)
    q, r = torch.randn(Q, D, generator=g).cuda(), torch.randn(N, D, generator=g).cuda()
    for _ in range(5):
        knn_float(r, q, k, _lib.WV_METRIC_IP)
    torch.cuda.synchronize()
    for stop in (1, 2, 3, 4, 5, 0):
        os.environ["WV_TK_STOP"] = str(stop)
        for _ in range(REPS):
            knn_float(r, q, k, _lib.WV_METRIC_IP)
        torch.cuda.synchronize()
        print(f"stop={stop}", flush=True)

#!/usr/bin/env python3
"""wv_knn_float timed alone (HIP events, clocks warmed up) at the shapes the reference's evaluator produces: the c1
database with real-valued 384-d embeddings (IP and L2, k = 5000 and small k), the c0 shape at k = N, a c3 shard.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import _lib  # noqa: E402
from wvhash.engine.get_knn import knn_float  # noqa: E402


def t_us(fn, reps=10, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def case(Q, N, D, k, metric, name):
    g = torch.Generator().manual_seed(7)
    q = torch.randn(Q, D, generator=g).cuda()
    r = torch.randn(N, D, generator=g).cuda()
    if "outliers" in name:                        # 1 % of the database rows 100 x longer: they stretch every row's score range
        r[torch.randperm(N, generator=g)[: N // 100].cuda()] *= 100.0
    us = t_us(lambda: knn_float(r, q, k, metric))
    fl = 2.0 * Q * N * D
    print(f"{name:34s} Q={Q:6d} N={N:6d} D={D:4d} k={k:6d}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (scores only counted)",
          flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    cases = [
        (2048, 25000, 384, 5000, _lib.WV_METRIC_IP, "c1 IP k=5000"),
        (2048, 25000, 384, 5000, _lib.WV_METRIC_L2, "c1 L2 k=5000"),
        (2048, 25000, 384, 100, _lib.WV_METRIC_L2, "c1 L2 k=100"),
        (2048, 25000, 384, 25000, _lib.WV_METRIC_IP, "c1 IP k=N"),
        (5823, 5717, 384, 5717, _lib.WV_METRIC_IP, "c0 IP k=N"),
        (5000, 117224, 384, 5000, _lib.WV_METRIC_L2, "c3 L2 k=5000"),
        (2048, 25000, 64, 5000, _lib.WV_METRIC_IP, "c1 IP k=5000 D=64 (tanh codes)"),
        (2048, 25000, 384, 5000, _lib.WV_METRIC_IP, "c1 IP k=5000 outliers"),
        (2048, 25000, 384, 5000, _lib.WV_METRIC_L2, "c1 L2 k=5000 outliers"),
    ]
    exact = any(only == c[5] for c in cases)
    for c in cases:
        if (only == c[5]) if exact else (only in c[5]):
            case(*c)

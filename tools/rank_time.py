#!/usr/bin/env python3
"""Ranking kernels, timed alone back to back (HIP events, clocks warmed up): c1 lists / lists + distances / fused mAP, the
8-GPU shard steps, c0 and c3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402


def t_us(fn, reps=20, warm=30):
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


def case(name, Q, N, nbits, k, lc=38, p=0.10):
    ql, rl = synth.multi_hot_labels(Q, lc, p, 1), synth.multi_hot_labels(N, lc, p, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    qlp, labels = H.pack_labels(ql.cuda()), H.PreparedLabels(H.pack_labels(rl.cuda()))
    out = [f"{name:28s} Q={Q:6d} N={N:6d} {nbits:3d}b k={k:6d}:"]
    out.append(f"lists {t_us(lambda: H.hamming_topk(qp, prep, nbits, k, want_dist=False)):7.1f} us")
    out.append(f"lists+dist {t_us(lambda: H.hamming_topk(qp, prep, nbits, k)):7.1f} us")
    if H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k) is not None:
        out.append(f"fused mAP {t_us(lambda: H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)):7.1f} us")
    if N <= H.SHARD_ROWS_MAX:
        out.append(f"hist {t_us(lambda: H.hamming_hist(qp, prep, nbits)):7.1f} us")
        kk = min(k, N, 1088)
        out.append(f"rows16[{kk}] {t_us(lambda: H.hamming_topk_rows16(qp, prep, nbits, kk)):7.1f} us")
        out.append(f"prefix[{kk}] {t_us(lambda: H.hamming_shard_prefix(qp, prep, nbits, kk)):7.1f} us")
        if labels.ok:
            out.append(f"relbits[{kk}] {t_us(lambda: H.hamming_shard_relbits(qp, prep, labels, qlp, nbits, kk)):7.1f} us")
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    case("c1", 2048, 25000, 64, 5000)
    case("c1 shard 1/8 (8x2048 queries)", 16384, 3125, 64, 3125)
    case("c1 shard 1/2 (2x2048 queries)", 4096, 12500, 64, 5000)
    case("c0 (VOC, 16 bit, k=N)", 5823, 5717, 16, 5717, lc=20, p=0.07)
    case("c3 shard 1/8", 5000, 14653, 128, 5000, lc=80, p=0.036)
    case("c1 k=N", 2048, 25000, 64, 25000)

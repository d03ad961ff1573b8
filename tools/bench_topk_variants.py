#!/usr/bin/env python3
"""Ranking kernel variants (WV_TOPK_V2 = 0: first-generation kernel, 256 / 64: windowed kernel with that many threads
per query) on the c1 shape and on the per-rank shape of an 8-way sharded search.  HIP-event timing, prepared database."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    shapes = [("c1", 2048, 25000, 64, 5000, False), ("c1 no dist row", 2048, 25000, 64, 5000, True),
              ("shard 1/8 of c1, all ranks' queries", 16384, 3125, 64, 3125, True),
              ("shard 1/2", 4096, 12500, 64, 5000, True),
              ("c3 shard 128 bit", 5000, 14653, 128, 5000, False), ("c0", 5823, 5717, 16, 5717, False)]
    for name, Q, N, nbits, k, cum in shapes:
        labels_q, labels_r = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
        q, r = synth.structured_codes(labels_q, nbits, 3, 4), synth.structured_codes(labels_r, nbits, 3, 5)
        qp = H.pack_codes(q.cuda())
        prep = H.PreparedDB(H.pack_codes(r.cuda()), nbits)
        ref = None
        for variant, qb in (("0", "1"), ("256", "1"), ("256", "8"), ("64", "1"), ("64", "8")):
            os.environ["WV_TOPK_V2"] = variant
            os.environ["WV_TOPK_QB"] = qb
            try:
                us = timeit(lambda: H.hamming_topk(qp, prep, nbits, k, want_dist=not cum, want_cum=cum))
            except Exception as e:  # noqa: BLE001
                print(f"{name}: variant {variant}: {e}")
                continue
            out = H.hamming_topk(qp, prep, nbits, k, want_dist=not cum, want_cum=cum)
            same = "" if ref is None else f"  identical to variant 0: {torch.equal(out[0], ref)}"
            ref = out[0] if ref is None else ref
            bytes_alg = (Q + N) * nbits // 8 + Q * k * (4 if cum else 5)
            print(f"{name} Q={Q} N={N} {nbits}b k={k}: variant {variant} qb {'max' if qb != '1' else '1'}: {us:7.1f} us  {bytes_alg / us / 1e3:7.1f} GB/s{same}", flush=True)
    os.environ.pop("WV_TOPK_V2", None)
    os.environ.pop("WV_TOPK_QB", None)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""wv_hamming_map_at_k (ranking + AP in one kernel, no lists) against wv_hamming_topk (lists only) + wv_map_at_k, c1 shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402


def timeit(fn, n=50):
    for _ in range(60):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, Q, N, nbits, k in [("c1", 2048, 25000, 64, 5000), ("c0", 5823, 5717, 16, 5717), ("c3 shard", 5000, 14653, 128, 5000)]:
    ql, rl = synth.multi_hot_labels(Q, 38, 0.10, 1), synth.multi_hot_labels(N, 38, 0.10, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    labels = H.PreparedLabels(rlp)

    def two():
        idx, _ = H.hamming_topk(qp, prep, nbits, k, want_dist=False)
        return H.map_at_k(idx, qlp, rlp)

    fused = H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)
    t2 = timeit(two)
    if fused is None:
        print(f"{name}: two kernels {t2:.1f} us; fused: not supported")
        continue
    t1 = timeit(lambda: H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k))
    same = torch.equal(fused[0], two()[0])
    print(f"{name} Q={Q} N={N} {nbits}b k={k}: ranking + AP kernels {t2:.1f} us, fused {t1:.1f} us, identical AP: {same}", flush=True)

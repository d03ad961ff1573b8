#!/usr/bin/env python3
"""c3 shapes on ONE GPU: the whole COCO-sized database (117,218 x 128 bit, 5,000 queries) ranked unsharded (first-generation
kernel: the windowed one takes databases below 32,768 rows), and the per-rank steps of the 8-way sharded search."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


Q, N, nbits = 5000, 117218, 128
ql, rl = synth.multi_hot_labels(Q, 80, 0.036, 1), synth.multi_hot_labels(N, 80, 0.036, 2)
q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
prep = H.PreparedDB(rp, nbits)
for k in (5000, 117218):
    us = timeit(lambda: H.hamming_topk(qp, rp, nbits, k, want_dist=False), reps=5)
    print(f"first-generation kernel, whole database: {Q} queries x {N} rows, {nbits} bit, k={k}: {us:9.0f} us  "
          f"({Q / us * 1e6:,.0f} queries/s)", flush=True)
us = timeit(lambda: H.hamming_topk(qp, prep, nbits, 5000, want_dist=False), reps=5)
same = torch.equal(H.hamming_topk(qp, prep, nbits, 5000)[0], H.hamming_topk(qp, rp, nbits, 5000)[0])
print(f"{len(prep.parts)} virtual shards of {prep.per} rows through the windowed kernel, k=5000: {us:9.0f} us  "
      f"({Q / us * 1e6:,.0f} queries/s), lists identical: {same}", flush=True)
qlp0, labs = H.pack_labels(ql.cuda()), H.PreparedLabels(H.pack_labels(rl.cuda()))
us = timeit(lambda: H.hamming_map_at_k(qp, prep, labs, qlp0, nbits, 5000), reps=5)
print(f"mAP@5000 through virtual shards + relevance strings: {us:9.0f} us", flush=True)
idx, _ = H.hamming_topk(qp, prep, nbits, 5000, want_dist=False)
qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
print(f"map_at_k (k=5000, 80 labels = 2 words): {timeit(lambda: H.map_at_k(idx, qlp, rlp)):.0f} us")
world = 8
per = (N + world - 1) // world
shard = H.PreparedDB(rp[:per].contiguous(), nbits)
qa = qp.repeat(world, 1)[: world * (Q // world + 1)].contiguous()          # every rank ranks all ranks' queries
t_h = timeit(lambda: H.hamming_hist(qa, shard, nbits))
send = 5000 // world * 2
t_r = timeit(lambda: H.hamming_topk_rows16(qa, shard, nbits, send))
print(f"8-way shard ({per} rows), {qa.shape[0]} queries: histograms {t_h:.0f} us + {send}-entry lists {t_r:.0f} us")

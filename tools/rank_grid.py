#!/usr/bin/env python3
"""c1 ranking launch time against the grid size of the persistent kernel (diagnostic build: WV_R2_GRID pins it)."""
import os
import sys

os.environ["WVHASH_DIAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from rank_time import t_us  # noqa: E402
from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402

Q, N, nbits, k = 2048, 25000, 64, 5000
ql, rl = synth.multi_hot_labels(Q, 38, 0.1, 1), synth.multi_hot_labels(N, 38, 0.1, 2)
q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
qlp, labels = H.pack_labels(ql.cuda()), H.PreparedLabels(H.pack_labels(rl.cuda()))
for grid in ("", "2048", "1536", "1280", "1024", "768", "683", "512", "256"):
    if grid:
        os.environ["WV_R2_GRID"] = grid
    print(f"grid {grid or 'auto':>5}: lists {t_us(lambda: H.hamming_topk(qp, prep, nbits, k, want_dist=False)):6.1f} us   "
          f"lists+dist {t_us(lambda: H.hamming_topk(qp, prep, nbits, k)):6.1f} us   "
          f"fused mAP {t_us(lambda: H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)):6.1f} us", flush=True)

#!/usr/bin/env python3
"""Is the ranking kernel's distance pass bound by L2 -> L1 traffic?  The same launch with every image load redirected to one
L1-resident 8 KB (variants -DWV_R2_SAMEROW; results wrong by construction), distance pass only (-DWV_R2_ABL=4) and whole kernel."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, os
sys.path.insert(0, os.path.join(sys.argv[1], "image-retrieval-wavelet_amd")); sys.path.insert(0, os.path.join(sys.argv[1], "tools"))
import torch
from rank_time import t_us
from wvhash import synth
from wvhash.engine import hamming as H
Q, N, nbits, k = 2048, 25000, 64, 5000
ql, rl = synth.multi_hot_labels(Q, 38, 0.1, 1), synth.multi_hot_labels(N, 38, 0.1, 2)
q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
print(f"lists {t_us(lambda: H.hamming_topk(qp, prep, nbits, k, want_dist=False)):6.1f} us   hist-only {t_us(lambda: H.hamming_hist(qp, prep, nbits)):6.1f} us")
"""
for name, lib in [("full kernel", "image-retrieval-wavelet_amd/wvhash/_lib/libwvhash_diag.so"), ("full kernel, L1-resident image", "tools/_variants/r2_samerow.so"),
                  ("distance pass only", "tools/_variants/r2_abl4.so"), ("distance pass only, L1-resident image", "tools/_variants/r2_abl4_samerow.so")]:
    for rep in range(2):
        env = dict(os.environ, WVHASH_LIB=os.path.join(ROOT, lib))
        out = subprocess.run([sys.executable, "-c", CODE, ROOT], env=env, capture_output=True, text=True)
        print(f"{name:40s}{out.stdout.strip()}{out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)

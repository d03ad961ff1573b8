#!/usr/bin/env python3
"""c1 ranking launch time with phases removed (variants built with -DWV_R2_ABL=n; the results are wrong by construction):
what each phase costs in the real launch, with the real overlap between workgroups."""
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
for Q, N, nbits, k in ((2048, 25000, 64, 5000), (16384, 3125, 64, 1088)):
    ql, rl = synth.multi_hot_labels(Q, 38, 0.1, 1), synth.multi_hot_labels(N, 38, 0.1, 2)
    q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    qp, prep = H.pack_codes(q.cuda()), H.PreparedDB(H.pack_codes(r.cuda()), nbits)
    print(f"  Q={Q} N={N} k={k}: lists {t_us(lambda: H.hamming_topk(qp, prep, nbits, k, want_dist=False)):6.1f} us", end="")
print()
"""
for name, lib in [("full kernel", "image-retrieval-wavelet_amd/wvhash/_lib/libwvhash_diag.so"), ("1: no list stores", "tools/_variants/r2_abl1.so"),
                  ("2: + no placement", "tools/_variants/r2_abl2.so"), ("3: + no counting", "tools/_variants/r2_abl3.so"),
                  ("4: distance pass only", "tools/_variants/r2_abl4.so"), ("5: full, placement stores conflict-free", "tools/_variants/r2_abl5.so"), ("full kernel again", "image-retrieval-wavelet_amd/wvhash/_lib/libwvhash_diag.so")]:
    env = dict(os.environ, WVHASH_LIB=os.path.join(ROOT, lib))
    out = subprocess.run([sys.executable, "-c", CODE, ROOT], env=env, capture_output=True, text=True)
    print(f"{name:42s}{out.stdout.strip()}{out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)

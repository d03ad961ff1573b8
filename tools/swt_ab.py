#!/usr/bin/env python3
"""A/B of libwvhash builds on the c1 SWT (db2 L3, 2048 planar uint8 images): launched back to back, reference and band-major
output, each library in a fresh process, alternating, twice.  usage: swt_ab.py lib1.so lib2.so ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, os
sys.path.insert(0, os.path.join(sys.argv[1], "image-retrieval-wavelet_amd"))
import torch
from wvhash.transforms import swt2d
Q = 2048
img = torch.randint(0, 256, (Q, 3, 224, 224), dtype=torch.uint8, device="cuda")
def t(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ref = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")
bm = torch.empty((4, Q, 3, 224, 224), dtype=torch.float32, device="cuda")
print(f"reference layout {t(lambda: swt2d(img, 'db2', 3, out=ref)):.4f} ms   band-major {t(lambda: swt2d(img, 'db2', 3, out=bm, band_major=True)):.4f} ms   "
      f"haar L1 {t(lambda: swt2d(img, 'haar', 1, out=ref)):.4f} ms")
"""
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, WVHASH_LIB=os.path.join(ROOT, lib))
        out = subprocess.run([sys.executable, "-c", CODE, ROOT], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):24s} {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)

#!/usr/bin/env python3
"""SWT time into each of 10 output buffers allocated one after the other (all alive), then again in reverse order."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.transforms import swt2d  # noqa: E402

Q = 2048


def timeit(out, img, n=12):
    for _ in range(3):
        swt2d(img, "db2", 3, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        swt2d(img, "db2", 3, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


img = torch.from_numpy(synth.natural_images(64, 224, 224, seed=0)).permute(0, 3, 1, 2).contiguous().repeat(Q // 64, 1, 1, 1).cuda()
bufs = [torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda") for _ in range(10)]
for rnd in range(2):
    order = range(10) if rnd == 0 else reversed(range(10))
    print("  ".join(f"#{i}:{timeit(bufs[i], img):.3f}" for i in order), flush=True)
print("addresses (GiB):", [round(b.data_ptr() / 2 ** 30, 2) for b in bufs])
img2 = img.clone()
print("input cloned (allocated last):", "  ".join(f"#{i}:{timeit(bufs[i], img2):.3f}" for i in (0, 1, 5, 9)))

#!/usr/bin/env python3
"""Does the SWT's write rate depend on HOW its 4.9 GB output buffer was allocated (page / fragment placement)?  The bench's
'timed alone' band-major row -- a buffer allocated late in the run -- keeps reading 3-8 % faster than the step's buffer."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.transforms import swt2d  # noqa: E402

Q = 2048
N = Q * 3 * 4 * 224 * 224


def timeit(out, img, n=20):
    for _ in range(5):
        swt2d(img, "db2", 3, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        swt2d(img, "db2", 3, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


first = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")          # before anything else
img = torch.from_numpy(synth.natural_images(64, 224, 224, seed=0)).permute(0, 3, 1, 2).contiguous().repeat(Q // 64, 1, 1, 1).cuda()
print(f"allocated first                         : {timeit(first, img):.4f} ms  ptr % 2MiB = {first.data_ptr() % (1 << 21)}", flush=True)
second = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")
print(f"allocated second (first still alive)    : {timeit(second, img):.4f} ms  ptr % 2MiB = {second.data_ptr() % (1 << 21)}", flush=True)
del first
third = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")           # reuses the cached block of `first`
print(f"reusing the freed first block           : {timeit(third, img):.4f} ms  ptr % 2MiB = {third.data_ptr() % (1 << 21)}", flush=True)
del second, third
torch.cuda.empty_cache()
fresh = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")
print(f"after empty_cache (fresh hipMalloc)     : {timeit(fresh, img):.4f} ms  ptr % 2MiB = {fresh.data_ptr() % (1 << 21)}", flush=True)
del fresh
torch.cuda.empty_cache()
big = torch.empty(N + (1 << 22), dtype=torch.float32, device="cuda")
for off in (0, 1024, 1 << 19, (1 << 19) + 256):
    view = big[off:off + N].view(Q, 3, 4, 224, 224)
    print(f"slice of a larger buffer, offset {off * 4:8d} B: {timeit(view, img):.4f} ms", flush=True)
print(f"band-major into the same large buffer   : ", end="")
bm = big[:N].view(4, Q, 3, 224, 224)
for _ in range(5):
    swt2d(img, "db2", 3, out=bm, band_major=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    swt2d(img, "db2", 3, out=bm, band_major=True)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 20:.4f} ms")

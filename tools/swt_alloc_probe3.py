#!/usr/bin/env python3
"""Inside ONE 12 GiB allocation: SWT time into 4.9 GB slices at offsets of 0 ... 6 GiB (is it the address or the allocation?),
for three such allocations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.transforms import swt2d  # noqa: E402

Q = 2048
N = Q * 3 * 4 * 224 * 224


def timeit(out, img, n=10):
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
keep = []
for a in range(3):
    big = torch.empty(3 * (1 << 30), dtype=torch.float32, device="cuda")      # 12 GiB
    keep.append(big)
    res = []
    for off_mib in (0, 64, 256, 1024, 2048, 3072, 4096, 6144):
        off = off_mib * (1 << 20) // 4
        res.append(f"+{off_mib}MiB:{timeit(big[off:off + N].view(Q, 3, 4, 224, 224), img):.3f}")
    print(f"allocation {a} @ {big.data_ptr() / 2 ** 30:.1f} GiB: " + "  ".join(res), flush=True)

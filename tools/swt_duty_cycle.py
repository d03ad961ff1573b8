#!/usr/bin/env python3
"""Is the SWT kernel's rate tied to what runs around it?  The same launch timed (HIP events around each launch) back to back,
with an idle gap after every launch, and with the MFMA-heavy head between launches (the step's duty cycle)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.models import get_fusion_head  # noqa: E402
from wvhash.transforms import swt2d  # noqa: E402

Q = 2048
img = torch.from_numpy(synth.natural_images(64, 224, 224, seed=0)).permute(0, 3, 1, 2).contiguous().repeat(Q // 64, 1, 1, 1).cuda()
out = torch.empty((Q, 3, 4, 224, 224), dtype=torch.float32, device="cuda")
head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4}, [384] * 4)
head.load_state_dict(synth.head_state(384, 4, "concat", 0))
head = head.cuda().eval()
feats = list(torch.stack(synth.band_features(Q, 384, 1)).cuda().unbind(0))


def run(between, n=40):
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        swt2d(img, "db2", 3, out=out)
        e1.record()
        evs.append((e0, e1))
        between()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) for a, b in evs]
    return sum(t[10:]) / len(t[10:]), min(t[10:])


with torch.no_grad():
    for name, fn in (("back to back", lambda: None), ("idle 0.4 ms after each", lambda: torch.cuda._sleep(900_000)),
                     ("idle 2 ms after each", lambda: torch.cuda._sleep(4_500_000)), ("head (0.25 ms, MFMA) after each", lambda: head(feats)),
                     ("back to back", lambda: None)):
        for _ in range(3):
            swt2d(img, "db2", 3, out=out)
        mean, best = run(fn)
        print(f"{name:34s}: SWT mean {mean:.4f} ms, best {best:.4f} ms", flush=True)

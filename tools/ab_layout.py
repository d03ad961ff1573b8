"""In-process A/B of the step with planar vs interleaved uint8 input (same box, alternating)."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import torch, bench
from wvhash.transforms import swt2d
p = bench.Pipeline(2048, 0, 1, torch.device("cuda", 0))
planar = p.images
inter = p.images.permute(0, 2, 3, 1).contiguous()
def run(img, cl, n=20):
    p.stage_swt = lambda: swt2d(img, bench.WAVELET, bench.LEVEL, channels_last=cl, out=p.bands)
    for _ in range(3): p.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): p.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for r in range(4):
    a = run(planar, False); b = run(inter, True)
    print(f"round {r}: planar {a:.4f} ms/step   interleaved {b:.4f} ms/step", flush=True)

import sys, os, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "image-retrieval-wavelet_amd"))
from wvhash.transforms import swt2d
def timeit(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
Q = 2048
img = torch.randint(0, 256, (Q, 224, 224, 3), dtype=torch.uint8, device="cuda")
imgc = img.permute(0, 3, 1, 2).contiguous()
imgf = (img.float() / 255)
for name, x, cl, od in [("u8 NHWC->f32", img, True, torch.float32), ("u8 NHWC->bf16", img, True, torch.bfloat16),
                        ("u8 NCHW->f32", imgc, False, torch.float32), ("f32 NHWC->f32", imgf, True, torch.float32)]:
    for wl in [("db2", 3), ("haar", 1)]:
        out = torch.empty((Q, 3, 4, 224, 224), dtype=od, device="cuda")
        ms = timeit(lambda: swt2d(x, wl[0], wl[1], channels_last=cl, out_dtype=od, out=out))
        inb = x.element_size() * 3 * 224 * 224; outb = out.element_size() * 12 * 224 * 224
        print(f"{name} {wl}: {ms:.3f} ms  {Q*(inb+outb)/ms/1e6:.0f} GB/s", flush=True)
for wl in [("db4", 1), ("bior4.4", 1), ("db2", 1), ("haar", 2)]:
    ms = timeit(lambda: swt2d(img, wl[0], wl[1], channels_last=True), 5)
    print(f"u8 NHWC->f32 {wl}: {ms:.3f} ms  {Q*(150528+2408448)/ms/1e6:.0f} GB/s (fallback kernels)", flush=True)

import sys, os, itertools
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "image-retrieval-wavelet_amd"))
import numpy as np, torch
from oracle import swt_np
from wvhash.transforms import swt2d
rng = np.random.default_rng(int(__import__("os").environ.get("WV_FUZZ_SEED", "2024")))
bad = 0; n = 0
wl_levels = [("haar", 1), ("haar", 2), ("haar", 3), ("db2", 1), ("db2", 2), ("db2", 3), ("db4", 1), ("bior4.4", 1), ("db4", 2), ("bior4.4", 2)]
for it in range(220):
    wl, lev = wl_levels[rng.integers(len(wl_levels))]
    m = 1 << lev
    H = int(rng.integers(1, 40)) * m if rng.random() < 0.7 else int(rng.choice([224, 256, 48, 64, 96, 40, 8]))
    W = int(rng.integers(1, 40)) * m if rng.random() < 0.7 else int(rng.choice([224, 256, 48, 64, 96, 40, 16, 32]))
    H -= H % m; W -= W % m
    if H < m or W < m: continue
    B = int(rng.integers(1, 5)); cl = bool(rng.integers(2)); fl = rng.random() < 0.25
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    x = torch.from_numpy(img)
    if fl: x = x.float() / 255.0
    if not cl: x = x.permute(0, 3, 1, 2).contiguous()
    try:
        y = swt2d(x.cuda(), wl, lev, channels_last=cl).cpu().numpy()
    except Exception as e:
        print("EXC", wl, lev, H, W, B, cl, fl, e); bad += 1; continue
    ref = swt_np.c_transform_batch(img, wl, lev)
    err = np.abs(y - ref).max(); n += 1
    if not (err <= 4e-6 * 2 ** lev):
        print("MISMATCH", wl, lev, H, W, B, cl, fl, err); bad += 1
print("cases", n, "bad", bad)

"""One-off fuzz of the ranking kernels against the oracle (ties, ragged sizes, select-then-sort path of the float k-NN)."""
import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "image-retrieval-wavelet_amd"))
import numpy as np, torch
from oracle import ranking
from wvhash import synth
from wvhash.engine import hamming as H, get_knn
rng = np.random.default_rng(int(__import__("os").environ.get("WV_FUZZ_SEED", "7")))
bad = n = 0
for it in range(60):                                   # Hamming top-k: random sizes
    Q = int(rng.integers(1, 40)); N = int(rng.integers(1, 9000)); nbits = int(rng.choice([16, 32, 48, 64, 128]))
    k = int(rng.integers(1, N + 1))
    q, r = synth.random_codes(Q, N, nbits, seed=int(rng.integers(1 << 30)))
    idx, d = H.hamming_topk(H.pack_codes(q.cuda()), H.pack_codes(r.cuda()), nbits, k)
    ri, rd = ranking.hamming_topk_stable(q, r, k)
    n += 1
    if not (torch.equal(idx.cpu().long(), ri.long()) and torch.equal(d.cpu().long(), rd.long())):
        print("HAMMING MISMATCH", Q, N, nbits, k); bad += 1
for it in range(60):                                   # float k-NN with heavy ties (small integer coordinates)
    Q = int(rng.integers(1, 30)); N = int(rng.integers(2, 6000)); D = int(rng.choice([4, 8, 64, 128]))
    k = int(rng.integers(1, N + 1)) if rng.random() < 0.5 else int(rng.integers(1, max(2, N // 3)))
    metric = str(rng.choice(["l2", "cosine"]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    lim = int(rng.choice([2, 3, 50]))
    q = torch.randint(-lim, lim + 1, (Q, D), generator=g).float(); r = torch.randint(-lim, lim + 1, (N, D), generator=g).float()
    if rng.random() < 0.3: r[N // 2:] = r[: N - N // 2].clone()          # exact duplicates -> ties across the selection threshold
    gi, gd = get_knn(r, q, k, False, distance_metric=metric)
    sd, si = ranking.knn_stable(r, q, k, metric)
    n += 1
    same = torch.equal(gi.cpu().long(), si.long())
    if not same:
        # integer-valued inputs: scores are exact in fp32 except for the sqrt of l2 -> compare scores, then ties by index
        ok = torch.allclose(gd.cpu(), sd, rtol=1e-6, atol=1e-6)
        print("KNN", "score-equal, order differs" if ok else "MISMATCH", metric, Q, N, D, k, lim); bad += 1
print("cases", n, "bad", bad)

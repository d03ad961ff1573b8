"""Fuzz of wv_knn_float: the one-kernel value-bin ranking (k_row_topk) against the radix kernel on the same scores -- equal bit
for bit -- over random shapes, k, metrics and score distributions (smooth, integer-valued, heavy-tailed, constant rows,
duplicates, few distinct values, sorted databases), and both against a stable torch sort of the scores the kernel itself
returns at k = N.  Needs the diagnostic library (WVHASH_DIAG=1 is set here)."""
import os
import sys

os.environ["WVHASH_DIAG"] = "1"
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "image-retrieval-wavelet_amd"))
import numpy as np, torch
from wvhash import _lib
from wvhash.engine.get_knn import knn_float

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
CASES = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for it in range(CASES):
    N = int(rng.choice([1, 2, 63, 64, 65, 1023, 1024, 1025, 4096, 4097, int(rng.integers(1, 5000)), int(rng.integers(5000, 60000))]))
    Q = int(rng.integers(1, 20))
    D = 4 * int(rng.integers(1, 17))
    k = int(rng.choice([1, N, min(N, 15360), min(N, 15361), int(rng.integers(1, N + 1)), int(rng.integers(1, min(N, 2000) + 1))]))
    metric = int(rng.choice([_lib.WV_METRIC_IP, _lib.WV_METRIC_L2, _lib.WV_METRIC_L2_SQUARED]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    kind = int(rng.integers(0, 7))
    q, r = torch.randn(Q, D, generator=g), torch.randn(N, D, generator=g)
    if kind == 1:                                   # integer-valued: ties en masse
        q, r = torch.randint(-2, 3, (Q, D), generator=g).float(), torch.randint(-2, 3, (N, D), generator=g).float()
    elif kind == 2:                                 # heavy tails: a few huge scores stretch the sampled range
        r = r * torch.exp(3 * torch.randn(N, 1, generator=g))
    elif kind == 3:                                 # duplicated database rows
        r[N // 2:] = r[: N - N // 2].clone()
    elif kind == 4:                                 # a constant row and a zero database block
        q[0] = 0
        r[: N // 3] = 0
    elif kind == 5:                                 # database sorted along one direction (scores correlate with the index)
        r = r[torch.argsort(r[:, 0])]
    elif kind == 6:                                 # few distinct rows
        r = r[torch.randint(0, min(N, 7), (N,), generator=g)]
    qc, rc_ = q.cuda(), r.cuda()
    os.environ.pop("WV_KNN_RADIX_ONLY", None)
    v1, i1 = knn_float(rc_, qc, k, metric)
    os.environ["WV_KNN_RADIX_ONLY"] = "1"
    v0, i0 = knn_float(rc_, qc, k, metric)
    os.environ.pop("WV_KNN_RADIX_ONLY")
    ok = torch.equal(i1, i0) and torch.equal(v1.view(torch.int32), v0.view(torch.int32))
    # ascending (value, index) for L2, descending value / ascending index for IP
    v, i = v1.cpu(), i1.cpu().long()
    sgn = -1.0 if metric == _lib.WV_METRIC_IP else 1.0
    if metric == _lib.WV_METRIC_L2:                 # ranked on the squared distance: the same lists as metric 2, values = its roots
        v2, i2 = knn_float(rc_, qc, k, _lib.WV_METRIC_L2_SQUARED)
        ok = ok and torch.equal(i2, i1) and torch.allclose(v2.sqrt(), v1, rtol=3e-7, atol=0)
        ok = ok and bool((v[:, 1:] >= v[:, :-1]).all())
    else:
        ok = ok and bool(((sgn * v[:, 1:] > sgn * v[:, :-1]) | ((v[:, 1:] == v[:, :-1]) & (i[:, 1:] > i[:, :-1]))).all())
    ok = ok and bool(((i >= 0) & (i < N)).all()) and all(len(set(row.tolist())) == k for row in i[:3])
    if not ok:
        bad += 1
        print(f"BAD case {it}: Q={Q} N={N} D={D} k={k} metric={metric} kind={kind}", flush=True)
    if it % 25 == 24:
        print(f"{it + 1} cases, {bad} bad", flush=True)
print(f"fuzz_knn: {CASES} cases, {bad} bad")
sys.exit(1 if bad else 0)

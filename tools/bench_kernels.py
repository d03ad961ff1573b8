#!/usr/bin/env python3
"""Kernel micro-benchmarks on the c1 shapes (HIP-event timing on the launch stream).
usage: python tools/bench_kernels.py [swt] [dist] [topk] [rankmap] [head] [map] [--q 2048] [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch  # noqa: E402

from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402
from wvhash.transforms import swt2d  # noqa: E402


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["swt", "dist", "topk", "head", "map"])
    ap.add_argument("--q", type=int, default=2048)
    ap.add_argument("--n", type=int, default=25000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--wavelet", default="db2")
    ap.add_argument("--level", type=int, default=3)
    a = ap.parse_args()
    Q, N = a.q, a.n
    if "swt" in a.what:
        nhwc = bool(os.environ.get("SWT_NHWC"))          # default: planar [Q,3,224,224] like bench.py
        img = torch.randint(0, 256, (Q, 224, 224, 3) if nhwc else (Q, 3, 224, 224), dtype=torch.uint8, device="cuda")
        for tile in os.environ.get("SWT_TILES", "default").split(";"):
            if tile != "default":
                os.environ["WV_SWT_TILE"] = tile
            ms = timeit(lambda: swt2d(img, a.wavelet, a.level, channels_last=nhwc), a.reps)
            gb = Q * (3 * 224 * 224 * (1 + 16)) / 1e9
            print(f"swt {a.wavelet} L{a.level} tile={tile}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s  {Q / ms * 1e3:.0f} img/s", flush=True)
    q, r = synth.random_codes(Q, N, 64, 0)
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    if "dist" in a.what:
        ms = timeit(lambda: H.hamming_dist(qp, rp), a.reps)
        print(f"hamming_dist: {ms * 1e3:.1f} us  {(Q * N + (Q + N) * 8) / ms / 1e6:.0f} GB/s", flush=True)
        prep = H.PreparedDB(rp, 64)
        out = torch.empty((Q, (N + 63) // 64 * 64), dtype=torch.uint8, device="cuda")
        import ctypes
        from wvhash import _lib
        lib = _lib.load()
        def raw():   # no allocation / Python wrapper in the loop: kernel launch cost only
            lib.wv_hamming_dist_prepared(_lib.ptr(qp), _lib.ptr(prep.blob), _lib.ptr(out), out.shape[1], Q, N, 1, _lib.stream_ptr())
        ms = timeit(raw, a.reps)
        print(f"hamming_dist_prepared (raw C call): {ms * 1e3:.1f} us  {(Q * N + (Q + N) * 8) / ms / 1e6:.0f} GB/s", flush=True)
    if "topk" in a.what:
        ws = H.TopkWorkspace()
        for k in (5000,):
            ms = timeit(lambda: H.hamming_topk(qp, rp, 64, k, workspace=ws), a.reps)
            print(f"hamming_topk k={k}: {ms * 1e3:.1f} us  {((Q + N) * 8 + Q * k * 5) / ms / 1e6:.0f} GB/s  {Q / ms * 1e3:.0f} q/s", flush=True)
    if "rankmap" in a.what:      # ranking + AP in one kernel (what the one-GPU step runs)
        ql = H.pack_labels(synth.multi_hot_labels(Q, 38, 0.1, 1).cuda())
        prep, labels = H.PreparedDB(rp, 64), H.PreparedLabels(H.pack_labels(synth.multi_hot_labels(N, 38, 0.1, 2).cuda()))
        ms = timeit(lambda: H.hamming_map_at_k(qp, prep, labels, ql, 64, 5000), a.reps)
        print(f"hamming_map_at_k k=5000: {ms * 1e3:.1f} us  {Q / ms * 1e3:.0f} q/s", flush=True)
    if "map" in a.what:
        idx, _ = H.hamming_topk(qp, rp, 64, 5000)
        ql = H.pack_labels(synth.multi_hot_labels(Q, 38, 0.1, 1).cuda())
        rl = H.pack_labels(synth.multi_hot_labels(N, 38, 0.1, 2).cuda())
        ms = timeit(lambda: H.map_at_k(idx, ql, rl), a.reps)
        print(f"map_at_k: {ms * 1e3:.1f} us", flush=True)
    if "knn" in a.what:
        from wvhash.engine import get_knn
        g = torch.Generator().manual_seed(0)
        for metric, D, k in (("l2", 384, 5000), ("cosine", 384, 5000), ("l2", 64, 100)):
            qe, re_ = torch.randn(Q, D, generator=g).cuda(), torch.randn(N, D, generator=g).cuda()
            ms = timeit(lambda: get_knn(re_, qe, k, False, distance_metric=metric), max(2, a.reps // 5))
            print(f"knn_float {metric} D={D} k={k}: {ms:.3f} ms  {Q / ms * 1e3:.0f} q/s", flush=True)
    if "head" in a.what:
        from wvhash.models import get_fusion_head
        head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4}, [384] * 4)
        head.load_state_dict(synth.head_state(384, 4, "concat", 0))
        head = head.cuda().eval()
        feats = list(torch.stack(synth.band_features(Q, 384, 1)).cuda().unbind(0))
        with torch.no_grad():
            ms = timeit(lambda: head(feats), a.reps)
        print(f"head: {ms * 1e3:.1f} us  {Q * 13.07e6 / ms / 1e9:.1f} TFLOP/s (13.07 MFLOP/sample executed by the one-launch front)", flush=True)


if __name__ == "__main__":
    main()

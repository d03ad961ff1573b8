#!/usr/bin/env python3
"""bench.py -- wavelet-hashing retrieval hot path on MI355X (contract: see the task brief).

Workload (BASELINE.json configs[1], "c1"): MIRFLICKR-25k shape, 3-level db2 SWT, 64-bit hash,
retrieval-only.  One STEP = one pass of the hot path over one batch of Q = 2048 synthetic query
images per GPU, everything resident in HBM when the timed region starts:

  images u8 [Q,3,224,224] (SURVEY 8(d) metric 1; the layout the deferred transform's DataLoader workers collate,
      custom_transforms.py) --wv_swt2d_forward(db2, L3)--> sub-bands f32 [Q,3,4,224,224]
  band CLS features f32 [4,Q,384] (synthetic: the DINOv2 backbone is outside the accelerated path,
      SURVEY.md 8 a-13) --wv_band_attn_pool (Nq=4, fp32 MFMA)--> [Q,384]
  --wv_hash_tail (hash_fc, BN, sign, pack)--> packed 64-bit query codes
  --wv_hamming_topk vs 25,000 packed database codes, k=5000--> ranked lists
  --wv_map_at_k--> AP per query -> mAP@5000

N > 1 (one process per GPU, RCCL): weak scaling -- every rank embeds its own Q images, the database
is row-sharded N ways; all_gather of the packed query codes, per-shard ranking of all N*Q queries,
all_to_all of the per-shard lists, GPU merge (wvhash/parallel.py).  value = N*Q / step time.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, live HIP-event timing),
"kernels" (every stage of the step, then rows "not in the step": the other BASELINE shapes -- c0, c3 -- and SURVEY 8(d)'s
grid of wavelets / batch sizes / head shapes, and the deferred SharedDinoHashing pipeline with a random-init ViT-S/14
consuming the sub-bands), "cpu_baseline" (the oracle = CPU port of the reference's op sequence, timed on this host's cores
on a bounded sample, N = 1 only).  N = 1 also reports "value_first_allocation": the same K steps timed BEFORE the two
setup steps that lift "value" (placement probe of the sub-band buffer, clock-ramp steps).  N > 1: config.exchange carries
the collectives counted per timed step and per-rank kernel / collective milliseconds.
"""
import argparse
import json
import os
import sys
import time
import traceback
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 measured copy)
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 (fp32 in/acc)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (the PyTorch ViT of the deferred-pipeline row; not a kernel of this library)

Q_PER_GPU, N_DB, NBITS, TOPK, N_CLASSES = 2048, 25000, 64, 5000, 38
H = W = 224
WAVELET, LEVEL = "db2", 3
EMBED, NQ, HEADS = 384, 4, 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clock-steps", type=int, default=40,
                    help="untimed steps run BEFORE the --warmup steps: after an idle period the compute-bound kernels keep "
                         "speeding up for ~25 steps (~35 ms) while the GPU's clocks ramp (kernel trace: head front 265 -> 212 us, "
                         "ranking 63 -> 49 us, the HBM-bound SWT unchanged); 0 = measure the ramp")
    ap.add_argument("--queries", type=int, default=Q_PER_GPU, help="query images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grid", action="store_true", help="skip the kernel rows that are not part of the step (c0 / c3 / SURVEY 8(d) grid)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("WV_BENCH_STREAMS", "0")), choices=(0, 1, 2),
                    help="2: the SWT of a batch runs on its own HIP stream, started when the head is done, beside hash tail, search "
                         "and collectives (stage pipelining); 1: everything on one stream; 0 (default): 1 on one GPU -- where the two "
                         "are equal within noise -- and 2 on several, where the exchange's latency hides behind the transform")
    ap.add_argument("--kernel-reps", type=int, default=10, help="launches per stage for the roofline timing")
    return ap.parse_args()


class Pipeline:
    def __init__(self, Q, rank, world, device, streams=1):
        from wvhash import synth
        from wvhash.engine import hamming as Hm
        from wvhash.models import get_fusion_head
        from wvhash.parallel import shard_bounds
        if os.environ.get("WV_BENCH_FAIL_SETUP_RANK") == str(rank):      # tests: a rank that dies during setup
            raise RuntimeError("injected setup failure (WV_BENCH_FAIL_SETUP_RANK)")
        self.Q, self.rank, self.world, self.dev = Q, rank, world, device
        self.swt_stream = torch.cuda.Stream(device=device) if streams == 2 else None
        self.band_major = os.environ.get("WV_BENCH_SWT_LAYOUT", "ref") == "band"       # A/B: the layout the models consume
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        self.images = torch.randint(0, 256, (Q, 3, H, W), generator=g, dtype=torch.uint8).to(device)
        # the sub-band buffer as it comes (tune_placement() may replace it later)
        self.bands, self.placement = self.place_swt_output(1)
        # the four backbones' CLS features, resident as slices of one [4, Q, E] buffer (what a pipeline that hands
        # each backbone an output slice produces): the head reads them in place
        self.feats = list(torch.stack(synth.band_features(Q, EMBED, seed=100 + rank)).to(device).unbind(0))
        self.head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": EMBED,
                                     "num_heads": HEADS, "num_queries": NQ, "sub_band_dropout_p": 0,
                                     "ortho_weight": 0.1}, [EMBED] * 4)
        self.head.load_state_dict(synth.head_state(EMBED, NQ, "concat", seed=0))
        self.head = self.head.to(device).eval()
        tail = synth.hash_tail_state(EMBED, NBITS, seed=1)
        self.hash_fc = torch.nn.Linear(EMBED, NBITS, bias=False)
        self.bn = torch.nn.BatchNorm1d(NBITS)
        self.hash_fc.load_state_dict({"weight": tail["hash_fc.weight"]})
        self.bn.load_state_dict({k[3:]: v for k, v in tail.items() if k.startswith("bn.")})
        self.hash_fc, self.bn = self.hash_fc.to(device).eval(), self.bn.to(device).eval()
        # database: label-correlated codes (same on every rank), this rank keeps rows [lo, hi)
        self.db_labels = synth.multi_hot_labels(N_DB, N_CLASSES, 0.10, seed=2)
        db_codes = synth.structured_codes(self.db_labels, NBITS, 3, 5)
        self.q_labels = synth.multi_hot_labels(Q, N_CLASSES, 0.10, seed=10 + rank)
        self.lo, self.hi, _ = shard_bounds(N_DB, world, rank)
        # the database is packed and laid out ONCE (wv_db_prepare), outside the timed region, like an index build
        self.db_packed_full = Hm.PreparedDB(Hm.pack_codes(db_codes.to(device)), NBITS)
        self.db_shard = (self.db_packed_full if world == 1 else
                         Hm.PreparedDB(self.db_packed_full.packed[self.lo:self.hi].contiguous(), NBITS))
        self.dblab = Hm.pack_labels(self.db_labels.to(device))
        self.qlab = Hm.pack_labels(self.q_labels.to(device))
        # one GPU: mAP needs no list -- ranking and AP run as one kernel (wv_hamming_map_at_k); with more ranks the
        # lists are what the shards exchange, so the two kernels stay
        self.lab_prepared = Hm.PreparedLabels(self.dblab) if world == 1 else None
        # more ranks: once the prefix length is known (first, exactly-sized list step) the shards send relevance strings
        # instead of lists (sharded_hamming_map_at_k)
        self.lab_shard = Hm.PreparedLabels(self.dblab[self.lo:self.hi].contiguous()) if world > 1 and self.hi > self.lo else None
        self.ws = Hm.TopkWorkspace()
        self.Hm = Hm
        self.db_codes_cpu = db_codes
        self.marks, self.marks_all = None, False
        # sharded search: prefix length of the list exchange.  None = sized exactly with one host read per step
        # (first warm-up step); afterwards the learned length + headroom, no host read, verified after the timed loop
        self.send_hint, self.needs, self.kin = None, [], min(TOPK, shard_bounds(N_DB, world, rank)[2])

    # -- the stages (each one C-ABI call) ---------------------------------------------------
    def place_swt_output(self, candidates):
        """Where the 4.9 GB sub-band buffer lies in HBM changes the SWT kernel's rate by +-3 % (first allocation of a
        process: up to +7 %; tools/swt_alloc_probe2.py: ten buffers allocated one after the other read 0.99 ... 1.05 ms,
        reproducibly per buffer): `candidates` buffers are allocated side by side, the kernel is timed on each (a few
        launches, HIP events) and the fastest one is kept -- setup, before any warm-up or timed step; the probe is reported
        in config.swt_output_placement.  WV_BENCH_SWT_CANDIDATES=1 takes the first allocation as it comes."""
        from wvhash.transforms import swt2d_place_output
        return swt2d_place_output(self.images, WAVELET, LEVEL, band_major=self.band_major, candidates=candidates,
                                  first=getattr(self, "bands", None))

    def tune_placement(self):
        """The setup step a long-running service does once: probe candidate allocations, keep the fastest.
        (Ranks rehearsing on ONE shared GPU over gloo keep the first allocation: their probes would time each other.)"""
        shared = self.world > 1 and os.environ.get("WV_DIST_BACKEND", "nccl") != "nccl"
        n = 1 if shared else int(os.environ.get("WV_BENCH_SWT_CANDIDATES", "8"))
        if n > 1:
            self.bands, self.placement = self.place_swt_output(n)

    def stage_swt(self):
        from wvhash.transforms import swt2d
        return swt2d(self.images, WAVELET, LEVEL, channels_last=False, out=self.bands, band_major=self.band_major)

    def stage_head(self):
        return self.head(self.feats)

    def stage_tail(self, fused):
        from wvhash.models import hash_tail
        return hash_tail(fused, self.hash_fc, self.bn, want=("packed",))["packed"]

    def stage_rank(self, packed):
        from wvhash.parallel import sharded_hamming_topk
        idx, d, need = sharded_hamming_topk(packed, self.db_shard, NBITS, TOPK, N_DB, workspace=self.ws,
                                            send_hint=self.send_hint, return_need=True, want_dist=False)
        if need is not None:
            self.needs.append(need)
        return idx, d

    def learn_send_hint(self):
        """After an exactly-sized step: exchange need + 12 % headroom from now on (rounded up to 64 entries)."""
        if self.world > 1 and self.needs:
            need = int(torch.stack([n.reshape(()) for n in self.needs]).max().item())
            self.send_hint = min(self.kin, (int(need * 1.12) + 63) // 64 * 64)
        self.needs = []

    def stage_map(self, idx):
        return self.Hm.map_at_k(idx, self.qlab, self.dblab)

    def _mark(self, name, stream=None):
        """HIP event on the stream the next / previous launch uses (only while a timing list is armed)."""
        if self.marks is not None and (self.marks_all or name.startswith("swt")):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream if stream is not None else torch.cuda.current_stream())
            self.marks.append((name, e))

    @torch.no_grad()
    def step(self):
        if self.swt_stream is not None:
            # stage pipelining: in the full system the backbone sits between the SWT and the head, so in steady
            # state the transform of one batch runs beside head/hash/ranking of the previous one.  The head goes FIRST: its
            # one-launch front takes 149 KB of LDS per CU and cannot share a CU with the transform's workgroups (2 x 58 KB) --
            # submitted after the persistent transform it would wait for all of it; submitted before, the transform starts
            # as the front's workgroups retire and everything behind the front (read-out, hash tail, search, collectives)
            # runs beside it.
            main = torch.cuda.current_stream()
            if self.marks_all:
                self._mark("step0")
            fused = self.stage_head()
            self._mark("head1")
            self.swt_stream.wait_stream(main)            # the transform starts when the head (and the previous step) is done
            with torch.cuda.stream(self.swt_stream):
                self._mark("swt0", self.swt_stream)
                bands = self.stage_swt()
                self._mark("swt1", self.swt_stream)
        else:
            self._mark("swt0")
            bands = self.stage_swt()
            self._mark("swt1")
            fused = self.stage_head()
            self._mark("head1")
        packed = self.stage_tail(fused)
        self._mark("tail1")
        fused_ap = (self.Hm.hamming_map_at_k(packed, self.db_shard, self.lab_prepared, self.qlab, NBITS, TOPK)
                    if self.lab_prepared is not None else None)
        sharded_ap = None
        if fused_ap is None and self.world > 1 and self.send_hint is not None:   # the same decision on every rank
            from wvhash.parallel import sharded_hamming_map_at_k
            sharded_ap = sharded_hamming_map_at_k(packed, self.qlab, self.db_shard, self.lab_shard, NBITS, TOPK, N_DB, self.send_hint)
        if fused_ap is not None:
            idx, ap = None, fused_ap[0]
            self._mark("rankmap1")
        elif sharded_ap is not None:
            idx, ap = None, sharded_ap[0]
            self.needs.append(sharded_ap[2])
            self._mark("rankmap1")
        else:
            idx, _ = self.stage_rank(packed)
            self._mark("rank1")
            ap, _ = self.stage_map(idx)
            self._mark("map1")
        if self.swt_stream is not None:
            torch.cuda.current_stream().wait_stream(self.swt_stream)
        return bands, packed, idx, ap


def time_stage(fn, reps):
    """Average device time of one call (HIP events on the launch stream), ms."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def stage_times_in_pipeline(p, reps):
    """Average device time of every stage inside full steps (HIP events between the stages, on the launch stream).
    A stage timed alone, back to back, can read differently: the SWT kernel run 10x in a row averages ~8 % slower
    than inside the step, where the head's compute-bound kernels give the write stream time to drain."""
    for _ in range(30):     # back to steady clocks first (the timed loop may lie an idle period behind us)
        p.step()
    p.marks, p.marks_all = [], True
    for _ in range(reps):
        p.step()
    torch.cuda.synchronize()
    marks, p.marks, p.marks_all = p.marks, None, False
    per = len(marks) // reps
    acc = {}
    for r in range(reps):
        m = marks[r * per:(r + 1) * per]
        if p.swt_stream is not None:
            # two streams: the stages overlap; report each stage from the event before it ON ITS OWN stream
            ev = dict(m)
            pairs = [("swt1", "swt0"), ("head1", "step0"), ("tail1", "head1"), ("rankmap1", "tail1"), ("rank1", "tail1"), ("map1", "rank1")]
            for name, prev in pairs:
                if name in ev and prev in ev:
                    acc[name] = acc.get(name, 0.0) + ev[prev].elapsed_time(ev[name])
            continue
        for (_, e0), (name, e1) in zip(m[:-1], m[1:]):
            acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
    return {k: v / reps for k, v in acc.items()}


def kernel_table(p, reps, swt_ms_live):
    Q = p.Q
    with torch.no_grad():
        fused = p.stage_head()
        packed = p.stage_tail(fused)
        st = stage_times_in_pipeline(p, reps)
        rows = []
        # algorithmic bytes / flops per launch (restated in DESIGN.md)
        swt_bytes = Q * (3 * H * W * 1 + 3 * 4 * H * W * 4)
        rows.append(("wv_swt2d_forward[k_swt_slide db2 L3 u8->f32]", "hbm", swt_bytes, swt_ms_live,
                     "HIP events around the launch in every timed step"))
        how = f"HIP events between the stages of {reps} extra full steps"
        # executed flops per sample: V 4x384x384, scores 4x384x32 (K projection folded into the query tokens), attention
        # out-projection 4x384x384, mlp 2 x 4x384x1536, read-out 1536x384 -- 13.07 MFLOP (the separate-launch path,
        # which projects K as well, executes 14.2)
        head_flops = 2 * (4 * EMBED * EMBED + 4 * EMBED * NQ * HEADS + NQ * EMBED * EMBED + 2 * NQ * EMBED * 4 * EMBED
                          + NQ * EMBED * EMBED)
        rows.append(("wv_band_attn_pool[3 launches: fused front k_head_front (V, scores, softmax, out-proj, LN, MLP; fp32 "
                     "MFMA) + read-out GEMM + LN]", "mfma", Q * head_flops, st["head1"], how))
        rows.append(("wv_hash_tail", "hbm", Q * (EMBED * 4 + 8) + NBITS * EMBED * 4, st["tail1"], how))
        if "rankmap1" in st:
            # ranking + AP in one kernel: codes, class-major label matrix and query labels in, one float per query out
            nw = (N_DB + 31) // 32
            rows.append(("wv_hamming_map_at_k[k_rank_window + AP in LDS, 64b N=25000 k=5000; no list leaves the CU: LDS / "
                         "latency-bound, the HBM fraction is not its yardstick]", "hbm",
                         (Q + N_DB) * NBITS // 8 + 64 * nw * 4 + Q * 8 + Q * 8, st["rankmap1"], how))
            idx_l = p.stage_rank(packed)[0]
            rows.append(("wv_hamming_topk[k_rank_window 64b N=25000 k=5000, lists only] (not in the step)", "hbm",
                         (Q + N_DB) * NBITS // 8 + Q * TOPK * 4, time_stage(lambda: p.stage_rank(packed), reps),
                         "timed alone, back to back"))
            rows.append(("wv_map_at_k (not in the step)", "hbm", Q * TOPK * 4 + (Q + N_DB) * 8 + Q * 8,
                         time_stage(lambda: p.stage_map(idx_l), reps), "timed alone, back to back"))
            del idx_l
        else:
            # the step ranks for mAP: lists only (4 bytes per entry), no distance row
            rows.append(("wv_hamming_topk[k_rank_window 64b N=25000 k=5000, lists only]", "hbm",
                         (Q + N_DB) * NBITS // 8 + Q * TOPK * 4, st["rank1"], how))
            rows.append(("wv_map_at_k", "hbm", Q * TOPK * 4 + (Q + N_DB) * 8 + Q * 8, st["map1"], how))
        from wvhash.transforms import swt2d
        bm = p.bands if p.band_major else torch.empty((4, Q, 3, H, W), dtype=torch.float32, device=p.dev)
        ref_buf = torch.empty((Q, 3, 4, H, W), dtype=torch.float32, device=p.dev) if p.band_major else p.bands
        rows.append(("wv_swt2d_forward_ex[same kernel, band-major output [4,Q,3,224,224]: what the models consume] "
                     "(not in the step)", "hbm", swt_bytes,
                     time_stage(lambda: swt2d(p.images, WAVELET, LEVEL, out=bm, band_major=True), reps),
                     "timed alone, back to back"))
        rows.append(("wv_swt2d_forward[same kernel, reference layout] (not in the step)", "hbm", swt_bytes,
                     time_stage(lambda: swt2d(p.images, WAVELET, LEVEL, out=ref_buf), reps), "timed alone, back to back"))
        del bm
        nhwc = p.images.permute(0, 2, 3, 1).contiguous()
        rows.append(("wv_swt2d_forward[same, interleaved [Q,224,224,3] input] (not in the step)", "hbm", swt_bytes,
                     time_stage(lambda: swt2d(nhwc, WAVELET, LEVEL, channels_last=True, out=ref_buf), reps),
                     "timed alone, back to back"))
        del nhwc
        rows.append(("wv_hamming_dist[k_hamming_dist u8 matrix]", "hbm", Q * N_DB + (Q + N_DB) * NBITS // 8,
                     time_stage(lambda: p.Hm.hamming_dist(packed, p.db_packed_full), reps),
                     "not part of the step; timed alone, back to back"))
        # the same kernel on the query set of an 8-GPU search (every rank ranks all 8 x 2048 queries): 410 MB per launch
        q8 = packed.repeat(8, 1).contiguous()
        rows.append(("wv_hamming_dist[same kernel, 16384 queries] (not in the step)", "hbm",
                     8 * Q * N_DB + (8 * Q + N_DB) * NBITS // 8,
                     time_stage(lambda: p.Hm.hamming_dist(q8, p.db_packed_full), reps), "timed alone, back to back"))
        del q8
    return finish_rows(rows)


def finish_rows(rows):
    out = []
    for name, bound, work, ms, how in rows:
        if bound == "hbm":
            ach, peak, unit = work / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = work / (ms * 1e-3) / 1e12, (BF16_MFMA_PEAK_TFLOPS if bound == "mfma_bf16" else F32_MFMA_PEAK_TFLOPS), "TFLOP/s"
            bound = "mfma"
        out.append({"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "ms": round(ms, 4), "work_per_launch": work, "timing": how})
    return out


def stream_ceilings(device):
    """What a plain streaming kernel reaches on THIS box (report the achievable ceiling beside the 8 TB/s
    spec): a 4 GiB float4 copy (read + write) and a 4 GiB fill (write only -- the SWT kernel is 94 % writes).  The buffers
    are far larger than the 256 MiB Infinity Cache: a 1 GiB fill repeated back to back reads 6.9 TB/s here because a
    quarter of it never leaves the cache."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.float32, device=device)
    b = torch.empty(n, dtype=torch.float32, device=device)
    copy_ms = time_stage(lambda: b.copy_(a), 5)
    fill_ms = time_stage(lambda: b.fill_(1.0), 5)
    del a, b
    return {"copy_ceiling_gbs": round(2 * n * 4 / (copy_ms * 1e-3) / 1e9, 1),
            "store_ceiling_gbs": round(n * 4 / (fill_ms * 1e-3) / 1e9, 1)}


def load_traffic(kernel_name):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (profiles/), if present."""
    path = os.path.join(ROOT, "profiles", "traffic_r03.json")
    if not os.path.exists(path):
        path = os.path.join(ROOT, "profiles", "traffic_r02.json")
    try:
        with open(path) as f:
            t = json.load(f)
        for key, val in t.items():
            if not key.startswith("_") and key in kernel_name:
                return val["hbm_bytes_per_launch"] if isinstance(val, dict) else val
    except (OSError, ValueError):
        pass
    return None


_POOL_WORKER = r"""
import sys, time
root, path, wl, lev, i, n, reps, t_start = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), \
    int(sys.argv[6]), int(sys.argv[7]), float(sys.argv[8])
sys.path.insert(0, root)
import numpy as np
from oracle import swt_np                      # numpy + ctypes only: no torch, no GPU in this process
imgs = np.ascontiguousarray(np.load(path, mmap_mode="r")[i::n])
swt_np.c_transform_batch(imgs[:1], wl, lev)
time.sleep(max(0.0, t_start - time.time()))    # all workers start together
for _ in range(reps):
    swt_np.c_transform_batch(imgs, wl, lev)    # one image, one channel at a time, like a DataLoader worker
print(time.time(), len(imgs) * reps)
"""


def cpu_swt_pool(imgs_hwc, nproc, reps):
    """The oracle's SWT on `nproc` single-threaded worker PROCESSES at once -- how the reference runs it
    (DataLoader(num_workers=...), evaluate.py:79-91).  Fresh interpreters (no fork of this GPU-owning process)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "imgs.npy")
        np.save(path, imgs_hwc)
        t_start = time.time() + 4.0
        procs = [subprocess.Popen([sys.executable, "-c", _POOL_WORKER, ROOT, path, WAVELET, str(LEVEL), str(i),
                                   str(nproc), str(reps), repr(t_start)], stdout=subprocess.PIPE, text=True)
                 for i in range(nproc)]
        outs = [pr.communicate()[0].split() for pr in procs]
    if any(pr.returncode for pr in procs):
        raise RuntimeError("cpu_baseline: an SWT worker process failed")
    done = sum(int(o[1]) for o in outs)
    wall = max(float(o[0]) for o in outs) - t_start
    return done / wall, done


def cpu_baseline(p):
    """The reference's op sequence on the host cores (oracle = CPU port), bounded sample."""
    from oracle import head_torch, ranking, swt_np
    from wvhash import synth
    # threads actually usable: the affinity mask, capped at the GPU box's 16-core share per GPU
    ncores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(ncores)
    n_img, n_q = min(512, p.Q), min(2048, p.Q)
    imgs = np.ascontiguousarray(p.images[:n_img].permute(0, 2, 3, 1).cpu().numpy())   # HWC, as PIL hands them over
    swt_np.c_transform_batch(imgs[:2], WAVELET, LEVEL)
    t0 = time.perf_counter()
    swt_np.c_transform_batch(imgs, WAVELET, LEVEL)       # per image, per channel, like the DataLoader worker
    t_swt1 = (time.perf_counter() - t0) / n_img
    reps = max(1, int(round(6.0 / (t_swt1 * n_img / ncores))))          # ~6 s of wall time for the pool
    pool_rate, pool_done = cpu_swt_pool(imgs, ncores, reps)
    t_swt = 1.0 / pool_rate
    sd = synth.head_state(EMBED, NQ, "concat", seed=0)
    feats = [f[:n_q].cpu() for f in p.feats]
    tail = synth.hash_tail_state(EMBED, NBITS, seed=1)
    with torch.no_grad():
        head_torch.band_attn_pool(feats, sd, HEADS)
        t0 = time.perf_counter()
        fused = head_torch.band_attn_pool(feats, sd, HEADS)
        codes = head_torch.hash_tail(fused, tail["hash_fc.weight"], tail["bn.weight"], tail["bn.bias"],
                                     tail["bn.running_mean"], tail["bn.running_var"])
        t_head = (time.perf_counter() - t0) / n_q
        codes[codes == 0] = 1.0
        ql = p.q_labels[:n_q]
        t0 = time.perf_counter()
        m_ref, ap_ref = ranking.calculate_maphashing(codes, ql, p.db_codes_cpu, p.db_labels, TOPK,
                                                     stable=True, return_per_query=True)
        t_rank = (time.perf_counter() - t0) / n_q
    # parity of the GPU path on the same sample (checker role of the oracle)
    with torch.no_grad():
        _, packed, idx, ap = p.step()
    ap_gpu = ap[:n_q].cpu().numpy()
    per_img = t_swt + t_head + t_rank
    return {
        "value": round(1.0 / per_img, 2), "unit": "query images/s", "cores": ncores, "kind": "port",
        "sample": f"SWT db2 L3: the C oracle in {ncores} single-threaded worker processes at once ({pool_done} images, "
                  f"per image/channel like DataLoader workers); {n_q} queries head/hash + the reference's per-query "
                  f"ranking loop (torch CPU, {ncores} threads), N_db={N_DB}, k={TOPK}; "
                  f"value = 1 / (sum of per-image stage times)",
        "ms_per_image": {"swt_pool": round(t_swt * 1e3, 4), "swt_one_process": round(t_swt1 * 1e3, 3),
                         "head_hash": round(t_head * 1e3, 4), "rank_map": round(t_rank * 1e3, 3)},
        "swt_images_per_s": {"one_process": round(1.0 / t_swt1, 1), f"{ncores}_processes": round(pool_rate, 1)},
        "map_at_k_cpu_sample": round(m_ref, 6),
        "max_abs_ap_diff_gpu_vs_cpu": float(np.abs(ap_gpu - np.asarray(ap_ref)).max()),
    }


def _row(name, bound, work, ms, how):
    return (name, bound, work, ms, how)


def extra_rows(p, reps):
    """Rows "not in the step": the other BASELINE shapes and SURVEY 8(d)'s grid, each with its algorithmic bytes / FLOPs
    (restated in DESIGN.md 4), timed alone back to back with HIP events on the launch stream."""
    from wvhash import synth
    from wvhash.engine import hamming as Hm
    from wvhash.engine.get_knn import knn_float
    from wvhash.models import SharedDinoHashing, get_fusion_head
    from wvhash.models.vit import vit_small_14
    from wvhash.transforms import swt2d
    from wvhash import _lib
    dev, rows, skipped, alone = p.dev, [], [], "not part of the step; timed alone, back to back"

    def add(name, bound, work, fn, nrep, how=alone):
        """One row; a shape the library refuses is listed under kernels_skipped instead of ending the run."""
        try:
            rows.append(_row(name, bound, work, time_stage(fn, nrep), how))
        except Exception as e:                                   # noqa: BLE001
            skipped.append(f"{name}: {type(e).__name__}: {str(e)[:200]}")
    img_bytes = lambda B, out_b=4: B * (3 * H * W + 12 * H * W * out_b)            # noqa: E731  u8 in, 4 bands out
    with torch.no_grad():
        # ---- SWT: the wavelets of the studies at level 1 (c0 = haar L1), c1's db2 L3 at smaller batches, bf16 output (c4)
        for wl, lev in (("haar", 1), ("db4", 1), ("bior4.4", 1)):
            add(f"wv_swt2d_forward[{wl} L{lev} u8->f32, B={p.Q}] (not in the step)", "hbm", img_bytes(p.Q),
                lambda: swt2d(p.images, wl, lev, out=p.bands if not p.band_major else None), reps)
        for B in (64, 256, 1024):
            if B < p.Q:
                x, o = p.images[:B], torch.empty((B, 3, 4, H, W), dtype=torch.float32, device=dev)
                add(f"wv_swt2d_forward[db2 L3 u8->f32, B={B}] (not in the step)", "hbm", img_bytes(B),
                    lambda: swt2d(x, WAVELET, LEVEL, out=o), reps)
        ob = torch.empty((p.Q, 3, 4, H, W), dtype=torch.bfloat16, device=dev)
        add(f"wv_swt2d_forward[db2 L3 u8->bf16 (c4), B={p.Q}] (not in the step)", "hbm", img_bytes(p.Q, 2),
            lambda: swt2d(p.images, WAVELET, LEVEL, out_dtype=torch.bfloat16, out=ob), reps)
        del ob
        # ---- ranking at c0 (VOC: 5,823 queries x 5,717 codes, 16 bit, k = N) and c3 (COCO: 5,000 x 117,218, 128 bit)
        for tag, Q, N, nbits, lc, pl, ks in (("c0", 5823, 5717, 16, 20, 0.07, (5717,)), ("c3", 5000, 117218, 128, 80, 0.036, (5000, 117218))):
            ql, rl = synth.multi_hot_labels(Q, lc, pl, 31), synth.multi_hot_labels(N, lc, pl, 32)
            q, r = synth.structured_codes(ql, nbits, 3, 33), synth.structured_codes(rl, nbits, 3, 34)
            qp, prep = Hm.pack_codes(q.to(dev)), Hm.PreparedDB(Hm.pack_codes(r.to(dev)), nbits)
            qlp, rlp = Hm.pack_labels(ql.to(dev)), Hm.pack_labels(rl.to(dev))
            labels = Hm.PreparedLabels(rlp)
            lw = rlp.shape[1]
            for k in ks:
                kk = "N" if k == N else str(k)
                code_b = (Q + N) * nbits // 8
                add(f"wv_hamming_topk[{tag}: {Q} x {N}, {nbits} bit, k={kk}, lists only] (not in the step)", "hbm",
                    code_b + Q * k * 4, lambda: Hm.hamming_topk(qp, prep, nbits, k, want_dist=False), max(2, reps // 2))
                fused = Hm.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)
                if fused is not None:
                    fn, how = (lambda: Hm.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)), "ranking + AP without lists (wv_hamming_map_at_k)"
                else:
                    def fn():
                        idx = Hm.hamming_topk(qp, prep, nbits, k, want_dist=False)[0]
                        return Hm.map_at_k(idx, qlp, rlp)
                    how = "wv_hamming_topk + wv_map_at_k (k beyond the fused kernel)"
                add(f"mAP@{kk} [{tag}: {Q} x {N}, {nbits} bit; {how}] (not in the step)", "hbm",
                    code_b + (Q + N) * 8 * lw + Q * 4, fn, max(2, reps // 2))
            del qp, prep, labels, qlp, rlp
        # ---- head: B in {256, 4096} x Nq in {1, 4, 8}
        for nq in (1, 4, 8):
            head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": EMBED, "num_heads": HEADS, "num_queries": nq,
                                    "sub_band_dropout_p": 0, "ortho_weight": 0.1}, [EMBED] * 4)
            head.load_state_dict(synth.head_state(EMBED, nq, "concat", seed=40 + nq))
            head = head.to(dev).eval()
            for B in (256, 4096):
                feats = list(torch.stack(synth.band_features(B, EMBED, seed=50 + nq)).to(dev).unbind(0))
                # separate launches (small batches) project K too: 14.2 MFLOP at Nq = 4; count what SURVEY 8(a-12) counts
                flops = 2 * (2 * 4 * EMBED * EMBED + 2 * HEADS * nq * 4 * (EMBED // HEADS) + nq * EMBED * EMBED
                             + 2 * nq * EMBED * 4 * EMBED + nq * EMBED * EMBED)
                add(f"wv_band_attn_pool[Nq={nq}, B={B}] (not in the step)", "mfma", B * flops, lambda: head(feats), reps)
        # ---- float k-NN (get_knn cosine / l2 path of non-hashing models): 2048 x 25,000, D = 384, k = 5000
        g = torch.Generator().manual_seed(7)
        qf, rf = torch.randn(p.Q, EMBED, generator=g).to(dev), torch.randn(N_DB, EMBED, generator=g).to(dev)
        add(f"wv_knn_float[{p.Q} x {N_DB}, D={EMBED}, IP, k={TOPK}: fp32 MFMA scores + one-kernel value-bin ranking] (not in the step)",
            "mfma", 2 * p.Q * N_DB * EMBED, lambda: knn_float(rf, qf, TOPK, _lib.WV_METRIC_IP), max(2, reps // 2))
        del qf, rf
        # ---- the pipeline with a consumer (SURVEY 8(d) metric 3, f-1): raw u8 batch -> SWT (bf16, band-major, kernel-written) ->
        # random-init ViT-S/14 on the 4 x B band images (stock PyTorch, bf16 autocast: the backbone is out of scope but it
        # READS the sub-bands) -> head -> hash -> packed codes -> ranking + AP.  main/engine/evaluate.py:26-64 is the sweep
        # it replaces; the reference's evaluate.py sets with_autocast = True.
        Bm = min(256, p.Q)
        net = SharedDinoHashing({"name": "dinov2_vits14", "frozen": True},
                                {"type": "cross_attention_advanced", "output_dim": EMBED, "num_heads": HEADS, "dropout": 0.1,
                                 "num_queries": NQ, "sub_band_dropout_p": 0.3, "ortho_weight": 0.1}, {"nbits": NBITS},
                                backbone=vit_small_14()).to(dev).eval()
        net.set_wavelet(LEVEL, WAVELET)
        raw, qlab = p.images[:Bm], p.qlab[:Bm]

        def model_step():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                packed = net.encode_packed(raw)
            return Hm.hamming_map_at_k(packed, p.db_packed_full, p.lab_prepared or Hm.PreparedLabels(p.dblab), qlab, NBITS, TOPK)

        try:
            ms = time_stage(model_step, 3)
            rows.append(_row(f"deferred SharedDinoHashing [raw u8 [{Bm},3,224,224] -> SWT db2 L3 (bf16) -> random-init ViT-S/14 (PyTorch, "
                             f"bf16 autocast, 4 x {Bm} images) -> head -> hash -> fused mAP@{TOPK}]: {Bm / (ms * 1e-3):.0f} query images/s "
                             "(not in the step; the backbone dominates and is outside the accelerated path)", "mfma_bf16",
                             # ViT-S/14 at 224: 257 tokens, 12 blocks of (4 E^2 attention + 8 E^2 MLP) MACs per token, + QK^T / AV
                             4 * Bm * 2 * (12 * (4 * EMBED * EMBED + 8 * EMBED * EMBED) * 257 + 12 * 2 * 257 * 257 * EMBED), ms,
                             "3 forward passes after 1 warm-up, HIP events; FLOPs = the ViT's, priced against the dense bf16 MFMA peak"))
        except Exception as e:                                   # noqa: BLE001
            skipped.append(f"deferred SharedDinoHashing pipeline: {type(e).__name__}: {str(e)[:200]}")
        del net
    return rows, skipped


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as fresh child
    processes (torch.distributed.run, one per GPU) and exit with their status.  This process has not touched
    the GPU (no HIP call, device_count() does not initialise it) and never does -- it only waits."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "WV_DIST_BACKEND" not in env and torch.cuda.device_count() < args.gpus:
        # fewer GPUs than ranks (a 1-GPU box): RCCL refuses two ranks on one device, so rehearse the N-rank path
        # with CPU-staged gloo collectives; the JSON line says so ("backend")
        env["WV_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_steps(p, steps, barrier):
    """EXACTLY `steps` steps between two barrier + synchronize pairs -> (seconds, last outputs, SWT ms per step (HIP events
    around the launch in every step), collectives counted during the steps, host synchronisations torch reported)."""
    from wvhash import parallel
    barrier()
    p.needs, p.marks = [], []          # one HIP event before and after the SWT launch of every timed step
    parallel.TRACE = parallel.ExchangeTrace()
    staged = p.world > 1 and dist.get_backend() != "nccl"      # gloo rehearsal: the host staging synchronises by design
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        if not staged:
            torch.cuda.set_sync_debug_mode("warn")              # any host read inside a step shows up as a warning
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = p.step()
        if not staged:
            torch.cuda.set_sync_debug_mode("default")
        barrier()
        elapsed = time.perf_counter() - t0
    trace, parallel.TRACE = parallel.TRACE, None
    syncs = None if staged else sum(1 for w in caught if "synchroniz" in str(w.message).lower())
    marks, p.marks = p.marks, None
    swt_ms = [a[1].elapsed_time(b[1]) for a, b in zip(marks[0::2], marks[1::2])]
    return elapsed, out, sum(swt_ms) / max(len(swt_ms), 1), trace, syncs


def per_rank_breakdown(p, reps=5):
    """A few extra steps with HIP events between the stages and around every collective: this rank's milliseconds per step."""
    from wvhash import parallel
    for _ in range(3):
        p.step()
    torch.cuda.synchronize()
    parallel.TRACE = parallel.ExchangeTrace(timing=True)
    p.marks, p.marks_all = [], True
    for _ in range(reps):
        p.step()
    torch.cuda.synchronize()
    marks, p.marks, p.marks_all = p.marks, None, False
    trace, parallel.TRACE = parallel.TRACE, None
    per = len(marks) // reps
    coll = {k: v / reps for k, v in trace.ms().items()}
    stages, step_ms = {}, 0.0
    for r in range(reps):
        m = marks[r * per:(r + 1) * per]
        if p.swt_stream is not None:                         # two streams: a stage is measured from the event before it on ITS stream
            ev = dict(m)
            for name, prev in (("swt1", "swt0"), ("head1", "step0"), ("tail1", "head1"), ("rankmap1", "tail1"), ("rank1", "tail1"),
                               ("map1", "rank1")):
                if name in ev and prev in ev:
                    stages[name] = stages.get(name, 0.0) + ev[prev].elapsed_time(ev[name]) / reps
            step_ms += max(ev["step0"].elapsed_time(ev["swt1"]), ev["step0"].elapsed_time(m[-1][1])) / reps
            continue
        step_ms += m[0][1].elapsed_time(m[-1][1]) / reps
        for (_, e0), (name, e1) in zip(m[:-1], m[1:]):
            stages[name] = stages.get(name, 0.0) + e0.elapsed_time(e1) / reps
    return {"rank": p.rank, "step_ms": round(step_ms, 4), "collectives_ms": {k: round(v, 4) for k, v in coll.items() if v},
            "kernels_ms": round(step_ms - (sum(coll.values()) if p.swt_stream is None else 0.0), 4),
            "note": ("two streams: the transform runs beside hash tail, search and collectives, so the stages overlap and step_ms "
                     "is the longer of the two streams") if p.swt_stream is not None else "one stream: step_ms = kernels_ms + collectives_ms",
            "stage_ms": {"swt": round(stages.get("swt1", 0.0), 4), "head": round(stages.get("head1", 0.0), 4),
                         "hash_tail": round(stages.get("tail1", 0.0), 4),
                         "search (kernels + collectives)": round(stages.get("rankmap1", stages.get("rank1", 0.0) + stages.get("map1", 0.0)), 4)},
            "bytes_sent_per_step": {k: v // reps for k, v in trace.bytes.items() if v}}


def run(args, rank, world, device):
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("WV_DIST_BACKEND", "nccl")   # "gloo": single-GPU rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = device if (world == 1 or backend == "nccl") else torch.device("cpu")   # where small control tensors live

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup; a rank that fails here tells the others instead of leaving them in their first collective
    p, setup_error = None, None
    streams = args.streams or (2 if world > 1 else 1)
    try:
        p = Pipeline(args.queries, rank, world, device, streams=streams)
    except Exception as e:                                   # noqa: BLE001
        setup_error = e
        print(f"[bench rank {rank}] setup failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    if world > 1:
        bad = torch.tensor([1 if setup_error is not None else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if setup_error is None:
                print(f"[bench rank {rank}] another rank failed during setup; exiting", file=sys.stderr, flush=True)
            raise SystemExit(1)
    elif setup_error is not None:
        raise setup_error

    out = p.step()                     # untimed: sizes the list exchange exactly (one host read), compiles nothing
    p.learn_send_hint()
    first = None
    if world == 1:
        # ---- as allocated, clocks as they are: W warm-up steps, K timed steps -- before the two setup steps below
        for _ in range(args.warmup):
            p.step()
        e_first, _, swt_first, _, _ = timed_steps(p, args.steps, barrier)
        first = {"value": round(args.queries * args.steps / e_first, 1), "ms_per_step": round(e_first / args.steps * 1e3, 4),
                 "swt_ms": round(swt_first, 4)}
    p.tune_placement()                 # setup: fastest of several candidate allocations of the sub-band buffer
    for _ in range(args.clock_steps):  # bring the clocks up (reported in config.clock_steps)
        p.step()
    for _ in range(args.warmup):
        out = p.step()
    elapsed, out, swt_ms_live, trace, host_syncs = timed_steps(p, args.steps, barrier)
    exchange = None
    if world > 1:
        from wvhash.parallel import exchange_ok
        per_step = {k: v / args.steps for k, v in trace.calls.items()}
        bad = torch.tensor([0 if exchange_ok(p.needs, p.send_hint, p.kin) else 1,
                            0 if per_step == {"all_gather": 1.0, "all_to_all": 1.0, "all_reduce": 0.0} else 1],
                           dtype=torch.int32, device=cdev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)             # every rank learns whether any rank fell short
        ok, lean = int(bad[0].item()) == 0, int(bad[1].item()) == 0
        # the sharded result of the last timed step against the unsharded kernel on THIS rank's queries and the whole
        # database (every rank holds it: 200 KB): the exchange over the real fabric must reproduce it bit for bit
        with torch.no_grad():
            ref = p.Hm.hamming_map_at_k(out[1], p.db_packed_full, p.Hm.PreparedLabels(p.dblab), p.qlab, NBITS, TOPK)
        same = torch.tensor([0 if (ref is not None and torch.equal(ref[0], out[3])) else 1], dtype=torch.int32, device=cdev)
        dist.all_reduce(same, op=dist.ReduceOp.MAX)
        matches = int(same.item()) == 0
        mine = per_rank_breakdown(p)
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        exchange = {"prefix_entries": p.send_hint, "max_possible": p.kin, "verified_exact": ok,
                    "ap_equals_unsharded_kernel_on_every_rank": matches,
                    "collectives_per_timed_step": {k: v for k, v in per_step.items()},
                    "one_all_gather_one_all_to_all_per_step": lean,
                    "bytes_sent_per_rank_per_timed_step": {k: v // args.steps for k, v in trace.bytes.items() if v},
                    "host_syncs_in_timed_steps": host_syncs,
                    "per_rank": ranks}
        if not ok:
            raise SystemExit(f"a timed step needed a longer list prefix than the hinted {p.send_hint}: the results of "
                             "that step are not exact, refusing to report a number")
        if not matches:
            raise SystemExit("the sharded search's average precisions differ from the unsharded kernel's on the same queries: "
                             "refusing to report a number")
    ap = out[3]
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    ap_sum = ap.double().sum().reshape(1).to(cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ap_sum)
    elapsed = float(t.item())
    total_q = args.queries * world
    map_at_k = float(ap_sum.item()) / total_q

    result = {
        "metric": "query images/sec + mAP@5000, MIRFLICKR-25k 64-bit hash, 1/2/4/8 MI355X",
        "value": round(total_q * args.steps / elapsed, 1),
        "unit": "query images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "map_at_5000": round(map_at_k, 6),
        "config": {
            "workload": "c1: MIRFLICKR-25k shape, retrieval-only: per GPU per step 2048 query images "
                        "3x224x224 u8 -> SWT db2 L3 -> band-attention head (Nq=4, E=384; DINOv2 backbone "
                        "out of scope, CLS features synthetic) -> 64-bit hash -> Hamming top-5000 vs 25,000 "
                        "codes -> mAP@5000",
            "queries_per_gpu": args.queries, "db_codes": N_DB, "nbits": NBITS, "k": TOPK,
            "wavelet": WAVELET, "level": LEVEL,
            "arithmetic": "fp32 SWT and head (fp32 MFMA), 64-bit popcount ranking, AP in fp32/fp64",
            "streams": streams,
            "clock_steps": args.clock_steps,   # untimed steps before the warm-up steps (GPU clock ramp after idle)
            "swt_output_placement": p.placement,  # setup: fastest of several candidate allocations of the sub-band buffer
            "host_syncs_in_timed_steps": host_syncs,
            "backend": (("rccl" if dist.get_backend() == "nccl" else
                         f"{dist.get_backend()} (REHEARSAL: {world} ranks share {torch.cuda.device_count()} GPU(s), "
                         "collectives staged through host memory)") if world > 1 else "none"),
            "exchange": exchange,
            "parallelism": (f"db row-sharded x{world}: all_gather(codes + label words), one ranking pass per shard, "
                            "all_to_all(relevance strings of the list prefixes + histograms), merge + AP on the receiving rank "
                            "(first step: lists, to size the prefix)") if world > 1 else "single GPU",
        },
    }
    if first is not None:
        # the same K steps before the two setup steps that lift `value` (placement probe, clock-ramp steps): what a caller
        # who allocates once and starts cold measures
        result["value_first_allocation"] = first["value"]
        result["first_allocation"] = {"ms_per_step": first["ms_per_step"], "swt_ms": first["swt_ms"],
                                      "what": f"{args.warmup} warm-up + {args.steps} timed steps on the sub-band buffer as first "
                                              "allocated, no clock-ramp steps; run before the tuned measurement"}
    if rank == 0 and world == 1:
        kt = kernel_table(p, args.kernel_reps, swt_ms_live)
        dom = max((k for k in kt if "not in the step" not in k["kernel"] and not k["kernel"].startswith("wv_hamming_dist")),
                  key=lambda k: k["ms"])
        result["roofline"] = {"kernel": dom["kernel"], "bound": dom["bound"], "achieved": dom["achieved"],
                              "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                              "traffic": load_traffic(dom["kernel"]),
                              "traffic_source": "rocprofv3 --pmc passes committed under profiles/ (not measured in this run)"}
        result["roofline"].update(stream_ceilings(device))
        if not args.no_grid:
            grid, skipped = extra_rows(p, args.kernel_reps)
            kt += finish_rows(grid)
            if skipped:
                result["kernels_skipped"] = skipped
        result["kernels"] = kt
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(p)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)   # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    code = 0
    try:
        run(args, rank, world, device)
    except SystemExit as e:
        code = e.code if isinstance(e.code, int) else 1
        if not isinstance(e.code, int) and e.code is not None:
            print(f"[bench rank {rank}] {e.code}", file=sys.stderr, flush=True)
    except BaseException:                                        # noqa: BLE001
        code = 1
        print(f"[bench rank {rank}] FAILED:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
    if world > 1 and code != 0:
        # Leave at once: the other ranks may sit in a collective this rank will never join, and tearing the process group
        # down would wait for them.  torch.distributed.run ends the whole group when one rank exits non-zero.
        sys.stdout.flush()
        os._exit(code)
    if world > 1 and dist.is_initialized():
        dist.destroy_process_group()
    raise SystemExit(code)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Single-GPU timing of the per-rank ranking work of the N-way sharded search (what one of N ranks does)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
import torch
from wvhash import synth
from wvhash.engine import hamming as H

def timeit(fn, reps=20):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

N_DB, K, QL = 25000, 5000, 2048
for world in (1, 2, 4, 8):
    per = (N_DB + world - 1) // world
    q, r = synth.random_codes(world * QL, per, 64, seed=world)
    qp = H.pack_codes(q.cuda()); prep = H.PreparedDB(H.pack_codes(r.cuda()), 64)
    kl = min(K, per)
    t_local = timeit(lambda: H.hamming_topk(qp, prep, 64, kl, want_dist=False, want_cum=True))
    send = min(kl, int(K / world * 1.6) + 64)
    t_hist = timeit(lambda: H.hamming_hist(qp, prep, 64))
    t_rows = timeit(lambda: H.hamming_topk_rows16(qp, prep, 64, send))
    t_one = timeit(lambda: H.hamming_shard_prefix(qp, prep, 64, send))       # hinted steady state: lists + histograms, one pass
    # what a rank receives for ITS QL queries: per shard the list prefix (16-bit local rows) + the cumulative histogram
    _, _, cum = H.hamming_topk(qp[:QL], prep, 64, kl, want_dist=False, want_cum=True)
    loc = torch.randint(0, per, (world, QL, send), dtype=torch.int32, device="cuda").to(torch.int16)
    cums = cum.unsqueeze(0).expand(world, QL, 66).contiguous()
    t_merge = timeit(lambda: H.topk_merge_cum(loc, cums, per, min(K, world * send), 64)) if world > 1 else 0.0
    # mAP without lists: relevance strings + histograms from the shard, merge + AP on the receiving rank
    lab = H.PreparedLabels(H.pack_labels(synth.multi_hot_labels(per, 38, 0.1, 2).cuda()))
    qlab = H.pack_labels(synth.multi_hot_labels(world * QL, 38, 0.1, 1).cuda())
    wbuf = torch.zeros((world * QL, H.relbits_wire_words(send, 64)), dtype=torch.int64, device="cuda")
    t_rel = timeit(lambda: H.hamming_shard_relbits(qp, prep, lab, qlab, 64, send, wire=wbuf))
    wires = H.hamming_shard_relbits(qp[:QL], prep, lab, qlab[:QL], 64, send).unsqueeze(0).expand(world, -1, -1).contiguous()
    t_mrel = timeit(lambda: H.merge_relbits_map(wires, send, min(K, world * send), 64)) if world > 1 else 0.0
    mb = world * QL * (send * 2 + 66 * 4) / 1e6
    print(f"world={world}: one-step local rank of {world*QL} queries vs {per} rows (k'={kl}): {t_local*1e3:.0f} us | "
          f"two-step: histograms {t_hist*1e3:.0f} us + {send}-entry 16-bit lists {t_rows*1e3:.0f} us | "
          f"hinted one pass (lists + histograms): {t_one*1e3:.0f} us | "
          f"compact merge {world} x {send}: {t_merge*1e3:.0f} us | relevance strings {t_rel*1e3:.0f} us + merge/AP {t_mrel*1e3:.0f} us "
          f"({world * QL * H.relbits_wire_words(send, 64) * 8 / 1e6:.1f} MB sent, one all_to_all) | lists: sent per rank ~{mb:.0f} MB "
          f"(int32+u8 lists: {world * QL * send * 5 / 1e6:.0f} MB)", flush=True)

#!/usr/bin/env python3
"""One-off randomized sweep of the kernels added in the second half of round 2:
  * head: one-launch front vs separate launches vs fp64 oracle (random batch, heads, queries);
  * wv_hamming_map_at_k vs wv_hamming_topk + wv_map_at_k (random shapes, label widths, tie-heavy codes);
  * sharded relevance strings (random shard counts / prefix lengths) vs the unsharded AP."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-retrieval-wavelet_amd")]
import torch  # noqa: E402

from oracle import head_torch  # noqa: E402
from wvhash import synth  # noqa: E402
from wvhash.engine import hamming as H  # noqa: E402
from wvhash.models import get_fusion_head  # noqa: E402

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0

# ---------------------------------------------------------------- head
for case in range(25):
    nq = rng.choice([4, 8])
    heads = rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 32] if nq == 4 else [1, 2, 3, 4, 6, 8, 12, 16])
    B = rng.choice([1, 2, 7, 8, 9, 31, 33, 100, 257, 1000, 1153, 2048, 3001])
    sd = synth.head_state(384, nq, "concat", seed=case)
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": nq, "num_heads": heads}, [384] * 4)
    head.load_state_dict(sd)
    head = head.cuda().eval()
    feats = synth.band_features(B, 384, seed=case + 100)
    dev = [f.cuda() for f in feats]
    out = {}
    with torch.no_grad():
        for front in ("1", "0"):
            os.environ["WV_HEAD_FRONT"] = front
            out[front] = head(dev).cpu()
    ref = head_torch.band_attn_pool(feats, sd, heads, dtype=torch.float64)
    e1, e0 = float((out["1"].double() - ref).abs().max()), float((out["0"].double() - ref).abs().max())
    ok = e1 < 5e-5 and e0 < 5e-5 and head._qproj_cache["blob"] is not None
    bad += not ok
    print(f"head nq={nq} heads={heads} B={B}: front {e1:.1e} separate {e0:.1e} {'ok' if ok else 'BAD'}", flush=True)
os.environ.pop("WV_HEAD_FRONT", None)

# ---------------------------------------------------------------- fused mAP, sharded relevance strings
for case in range(40):
    nbits = rng.choice([16, 32, 48, 64, 96, 128])
    N = rng.choice([40, 257, 1000, 3125, 4096, 4097, 12500, 25000, 32768])
    Q = rng.choice([1, 5, 64, 300, 4100])
    k = rng.randint(1, min(N, 8192))
    Lc = rng.choice([1, 5, 24, 38, 64, 65, 80, 128])
    ql, rl = synth.multi_hot_labels(Q, Lc, 0.1, case), synth.multi_hot_labels(N, Lc, 0.1, case + 1)
    if rng.random() < 0.5:
        q, r = synth.structured_codes(ql, nbits, 3, 4), synth.structured_codes(rl, nbits, 3, 5)
    else:   # few distinct codes: heavy ties
        g = torch.Generator().manual_seed(case)
        base = torch.randint(0, 2, (7, nbits), generator=g).float() * 2 - 1
        q, r = base[torch.randint(0, 7, (Q,), generator=g)], base[torch.randint(0, 7, (N,), generator=g)]
    qp, rp = H.pack_codes(q.cuda()), H.pack_codes(r.cuda())
    qlp, rlp = H.pack_labels(ql.cuda()), H.pack_labels(rl.cuda())
    idx, _ = H.hamming_topk(qp, rp, nbits, k, want_dist=False)
    ap_ref, nrel_ref = H.map_at_k(idx, qlp, rlp)
    prep, labels = H.PreparedDB(rp, nbits), H.PreparedLabels(rlp)
    fused = H.hamming_map_at_k(qp, prep, labels, qlp, nbits, k)
    if fused is None:
        msg = "fused: not taken"
        ok = True
    else:
        ok = torch.equal(fused[1], nrel_ref) and float((fused[0] - ap_ref).abs().max()) <= 1.2e-7
        msg = f"fused max|dAP| {float((fused[0] - ap_ref).abs().max()):.1e}"
    # shards
    G = rng.choice([2, 3, 5, 8])
    per = (N + G - 1) // G
    shards = []
    for g_ in range(G):
        lo, hi = min(N, g_ * per), min(N, (g_ + 1) * per)
        shards.append((lo, hi, H.PreparedDB(rp[lo:hi].contiguous(), nbits) if hi > lo else None,
                       H.PreparedLabels(rlp[lo:hi].contiguous()) if hi > lo else None))
    cums = torch.stack([H.hamming_hist(qp, db, nbits) if db is not None else
                        torch.zeros((Q, nbits + 2), dtype=torch.int32, device="cuda") for _, _, db, _ in shards])
    T = (cums.sum(0)[:, 1:] >= k).int().argmax(dim=1)
    need = int(torch.gather(cums, 2, (T + 1).view(1, Q, 1).expand(G, Q, 1).long()).max().item())
    send = min(min(k, per), need + rng.randint(0, 70))
    smsg = "sharded: not taken"
    if labels.ok and all(l is None or l.ok for *_, l in shards) and k <= 8192:
        wires = []
        for lo, hi, db, lab in shards:
            wire = torch.zeros((Q, H.relbits_wire_words(send, nbits)), dtype=torch.int64, device="cuda")
            if db is not None:
                H.hamming_shard_relbits(qp, db, lab, qlp, nbits, min(send, hi - lo), wire=wire, kin=send)
            wires.append(wire)
        owed = torch.zeros(1, dtype=torch.int32, device="cuda")
        ap, nrel = H.merge_relbits_map(torch.stack(wires), send, k, nbits, need_out=owed)
        sok = int(owed.item()) == need and torch.equal(nrel, nrel_ref) and torch.equal(ap, ap_ref)
        ok = ok and sok
        smsg = f"sharded G={G} send={send} need={need}: {'equal' if sok else 'DIFFERENT'}"
    bad += not ok
    print(f"map Q={Q} N={N} {nbits}b k={k} Lc={Lc}: {msg}; {smsg} {'ok' if ok else 'BAD'}", flush=True)
print(f"{bad} bad cases")
sys.exit(1 if bad else 0)

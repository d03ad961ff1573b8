"""A/B of the band-attention head: one-launch front (prepared weights) vs the separate launches, same inputs.
Prints max |difference|, both against the fp64 oracle, and HIP-event timings at the bench shape."""
import sys
import torch
sys.path[:0] = [".", "image-retrieval-wavelet_amd"]
from oracle import head_torch
from wvhash import synth
from wvhash.models import get_fusion_head
from wvhash.models.fusion import band_attn_pool


def run(head, feats, prepared):
    cache = head._qproj_cache if prepared else None
    return band_attn_pool(feats, head.effective_queries(), head.attn, head.norm1, head.norm2, head.mlp[0], head.mlp[2],
                          head.out_proj, pool_mean=False, qproj_cache=cache, qproj_key=head._query_key() if prepared else None)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for nq, heads, B in [(4, 8, 2048), (8, 8, 512), (4, 12, 100), (4, 8, 3), (8, 6, 37)]:
    sd = synth.head_state(384, nq, "concat", seed=nq + heads)
    head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": nq, "num_heads": heads}, [384] * 4)
    head.load_state_dict(sd)
    head = head.cuda().eval()
    feats_cpu = synth.band_features(B, 384, seed=B)
    feats = [f.cuda() for f in feats_cpu]
    with torch.no_grad():
        a = run(head, feats, True)
        b = run(head, feats, False)
        ref = head_torch.band_attn_pool(feats_cpu, sd, heads, dtype=torch.float64)
        print(f"Nq={nq} heads={heads} B={B}: |fused-separate|={float((a - b).abs().max()):.2e} "
              f"|fused-ref64|={float((a.cpu().double() - ref).abs().max()):.2e} |separate-ref64|={float((b.cpu().double() - ref).abs().max()):.2e}")
        if B >= 512:
            print(f"   fused {timeit(lambda: run(head, feats, True)):.1f} us   separate {timeit(lambda: run(head, feats, False)):.1f} us")

print("batch sweep (Nq=4, heads=8): fused vs separate, us")
sd = synth.head_state(384, 4, "concat", seed=1)
head = get_fusion_head({"type": "cross_attention_advanced", "output_dim": 384, "num_queries": 4, "num_heads": 8}, [384] * 4)
head.load_state_dict(sd)
head = head.cuda().eval()
with torch.no_grad():
    for B in (32, 64, 128, 256, 512, 1024, 1536, 2048, 4096):
        feats = [f.cuda() for f in synth.band_features(B, 384, seed=B)]
        print(f"   B={B}: fused {timeit(lambda: run(head, feats, True)):.1f}   separate {timeit(lambda: run(head, feats, False)):.1f}")

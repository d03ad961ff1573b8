// Average precision from relevance bits in list order -- the arithmetic and summation order of k_map_at_k (topk.hip):
// position p = round * TPQ + t, the j-th hit adds the fp32 quotient j / (p + 1) to a double, threads summed per wave,
// waves in index order.  Shared by the fused ranking + AP kernel (rank2.hip) and the merge of per-shard relevance
// strings (topk.hip).
#pragma once
#include "common.hpp"

namespace wv {

constexpr int kApRounds = 32;                // list positions per thread kept as bits: k <= 32 * TPQ

// dwords of LDS scratch ap_finish needs: hit counts [kApRounds][NW] + NW doubles (8-byte aligned)
template <int TPQ>
__host__ __device__ constexpr int ap_scratch_dwords() { return kApRounds * (TPQ / 64) + 2 * (TPQ / 64) + 2; }

// relbits: bit r = relevance of position r * TPQ + t (r < R <= 32).  cnt[r * NW + wave] = hits of that wave in round r,
// written by lane 0 of every wave BEFORE the call (the function starts with the group barrier that publishes them).
// SYNC: the barrier of the TPQ threads that share the list.
template <int TPQ, typename SYNC>
__device__ __forceinline__ void ap_finish(uint32_t relbits, uint32_t *scratch, int R, int t, float *__restrict__ ap_out,
                                          int32_t *__restrict__ nrel_out, SYNC group_barrier)
{
    constexpr int NW = TPQ / 64, CH = 8;
    const int lane = t & 63, wv = t >> 6;
    const uint32_t *cnt = scratch;
    double *wsum = reinterpret_cast<double *>(scratch + kApRounds * NW + (kApRounds * NW & 1));
    group_barrier();
    uint32_t running = 0;
    double acc = 0.0;
    for (int r0 = 0; r0 < R; r0 += CH) {
        uint32_t c[CH][NW];
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) c[u][w2] = cnt[min(r0 + u, R - 1) * NW + w2];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int r = r0 + u;
            if (r < R) {                                          // uniform
                uint32_t before = running, tot = 0;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    before += w2 < wv ? c[u][w2] : 0u;
                    tot += c[u][w2];
                }
                const bool rel = (relbits >> r) & 1u;
                const uint64_t m = __ballot(rel);
                if (rel) {
                    const uint32_t j = before + (uint32_t)mbcnt(m) + 1;
                    acc += (double)((float)j / (float)(r * TPQ + t + 1));
                }
                running += tot;
            }
        }
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) wsum[wv] = acc;
    group_barrier();
    if (t == 0) {
        double s = wsum[0];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) s += wsum[w2];
        *ap_out = running ? (float)(s / (double)running) : 0.0f;
        if (nrel_out) *nrel_out = (int32_t)running;
    }
}

}  // namespace wv

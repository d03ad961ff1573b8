// Average precision from relevance bits in list order -- the arithmetic and summation order of k_map_at_k (topk.hip):
// position p = round * TPQ + t, the j-th hit adds the fp32 quotient j / (p + 1) to a double, threads summed per wave,
// waves in index order.  Shared by the fused ranking + AP kernel (rank2.hip) and the merge of per-shard relevance
// strings (topk.hip).
#pragma once
#include "common.hpp"

namespace wv {

constexpr int kApRounds = 32;                // list positions per thread kept as bits at a time (one chunk of the walk)

// dwords of LDS scratch ap_finish needs: hit counts [kApRounds][NW] + NW doubles (8-byte aligned)
template <int TPQ>
__host__ __device__ constexpr int ap_scratch_dwords() { return kApRounds * (TPQ / 64) + 2 * (TPQ / 64) + 2; }

// The walk comes in chunks of up to kApRounds rounds, so that lists of any length go through the same arithmetic:
//   ApState st;  for every chunk { lane 0 of every wave writes cnt[r_local * NW + wave] = hits of that wave in the round;
//                                  group barrier;  ap_accum(relbits, cnt, rounds, first round, t, st);  group barrier; }
//   ap_final(st, ...).
// A thread's quotients are added to its double in increasing list position whatever the chunking: the result equals
// k_map_at_k's bit for bit.
struct ApState {
    uint32_t running = 0;    // hits before the current round, all threads
    double acc = 0.0;        // this thread's sum of fp32 quotients j / rank
};

// relbits: bit r = relevance of position (r_base + r) * TPQ + t (r < Rc <= 32); cnt[r * NW + wave] as above, already published.
template <int TPQ>
__device__ __forceinline__ void ap_accum(uint32_t relbits, const uint32_t *cnt, int Rc, int r_base, int t, ApState &st)
{
    constexpr int NW = TPQ / 64, CH = 8;
    const int wv = t >> 6;
    for (int r0 = 0; r0 < Rc; r0 += CH) {
        uint32_t c[CH][NW];
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) c[u][w2] = cnt[min(r0 + u, Rc - 1) * NW + w2];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int r = r0 + u;
            if (r < Rc) {                                         // uniform
                uint32_t before = st.running, tot = 0;
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
                    before += w2 < wv ? c[u][w2] : 0u;
                    tot += c[u][w2];
                }
                const bool rel = (relbits >> r) & 1u;
                const uint64_t m = __ballot(rel);
                if (rel) {
                    const uint32_t j = before + (uint32_t)mbcnt(m) + 1;
                    st.acc += (double)((float)j / (float)((r_base + r) * TPQ + t + 1));
                }
                st.running += tot;
            }
        }
    }
}

// wsum: NW doubles of LDS (8-byte aligned) that nothing else uses until the second barrier
template <int TPQ, typename SYNC>
__device__ __forceinline__ void ap_final(const ApState &st, double *wsum, int t, float *__restrict__ ap_out,
                                         int32_t *__restrict__ nrel_out, SYNC group_barrier)
{
    constexpr int NW = TPQ / 64;
    const int lane = t & 63, wv = t >> 6;
    const double acc = wave_sum_f64(st.acc);
    if (lane == 0) wsum[wv] = acc;
    group_barrier();
    if (t == 0) {
        double s = wsum[0];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) s += wsum[w2];
        *ap_out = st.running ? (float)(s / (double)st.running) : 0.0f;
        if (nrel_out) *nrel_out = (int32_t)st.running;
    }
}

// One chunk is the whole list (R <= kApRounds rounds): relbits: bit r = relevance of position r * TPQ + t.  cnt[r * NW + wave] =
// hits of that wave in round r, written by lane 0 of every wave BEFORE the call (the function starts with the group barrier that
// publishes them).  SYNC: the barrier of the TPQ threads that share the list.
template <int TPQ, typename SYNC>
__device__ __forceinline__ void ap_finish(uint32_t relbits, uint32_t *scratch, int R, int t, float *__restrict__ ap_out,
                                          int32_t *__restrict__ nrel_out, SYNC group_barrier)
{
    constexpr int NW = TPQ / 64;
    double *wsum = reinterpret_cast<double *>(scratch + kApRounds * NW + (kApRounds * NW & 1));
    group_barrier();
    ApState st;
    ap_accum<TPQ>(relbits, scratch, R, 0, t, st);
    ap_final<TPQ>(st, wsum, t, ap_out, nrel_out, group_barrier);
}

}  // namespace wv

// Windowed, branch-free Hamming ranking kernel for gfx950 (second generation of topk.hip's counting sort).
//
// Same contract as k_hamming_topk (topk.hip): the k nearest database codes of every query in ascending
// (distance, database index) order -- torch.argsort(stable=True) of accuracy_calculator.py:219-223 -- as an exact
// counting sort: thread t owns the contiguous item range [t*C, (t+1)*C) and a private column of an LDS count table.
// What changed, and why (rocprofv3 / ISA of the first kernel: 336 exec-mask branches per pass, 34 KB of LDS per
// query, 122 VGPRs, 1.7x write amplification from scattered 4-byte index stores):
//   * the table covers a WINDOW of 32 distance bins starting at the query's smallest distance, not all nbits+1 bins:
//     17 KB instead of 34 KB (distances of a query concentrate in far fewer than 32 bins; if the k-th neighbour lies
//     beyond the window the window slides on and the pass repeats -- exact for any input, one pass in practice);
//   * nothing in the per-item loops is conditional: items outside the window (and the padding items of the last
//     thread) go to a dummy table row, list entries beyond k to a per-lane trash slot;
//   * the ranked list is assembled in LDS as 16-bit item numbers and leaves with 16-byte coalesced stores
//     (global index = item + idx_offset), the distance row is regenerated from the bin boundaries, 16 bytes per store;
//   * TPQ = 64 runs one query per WAVE (no workgroup barrier at all, 4 queries per workgroup): the shape of a
//     row-sharded search, many queries against few rows each.
//   * a workgroup's time is a chain of latencies (phase stamps), so loads and returning LDS adds are kept
//     deep in flight: a lane loads 16 bytes of the database image (two 64-bit codes) per instruction, eight loads per
//     batch, sixteen returning adds before the first rank is used.  The kernel can compute the distances of QB queries
//     from one pass over the image (template parameter; the queries are then ranked one after the other through the same
//     LDS) -- measured slower than QB = 1 at every shape, so only QB = 1 is instantiated;
//   * k = 0 is a histogram-only mode (one count pass over all bins), rows16 output writes 16-bit local row numbers:
//     the two steps of the sharded search (wv_hamming_hist, wv_hamming_topk_rows16).
//   * optional: average precision of the list without ever writing it (wv_hamming_map_at_k).  calculate_maphashing
//     (accuracy_calculator.py:183-231) returns one number per query; the list, 20 KB per query, was written to HBM only
//     to be read back by the AP kernel, which then gathers 8-byte labels at random (the gather unit serves ~1 lane per
//     clock: half of that kernel's time).  Here the labels come as a class-major bit matrix: the OR of the rows of the
//     query's classes is its relevance bitmap (N bits, in LDS), and after placement the list in LDS is walked against
//     the bitmap -- same thread <-> position mapping and summation order as k_map_at_k, bit-identical AP.
// Covers databases (shards) of at most 32,768 rows -- C <= 128 items per thread, their distances cached in registers as
// bytes; 16-bit item numbers and counters -- and k small enough for the LDS list; everything else stays on topk.hip's kernel.
#include "common.hpp"
#include "ap_walk.hpp"

namespace wv {

// Diagnostic build only (-DWV_RANK2_STAMPS, tools/build_variant.sh): cycles per phase, summed over wave 0 of every query
// group into a device array that nothing else reads.
#ifdef WV_RANK2_STAMPS
__device__ unsigned long long g_rank2_stamps[8];
#define R2_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); if (t == 0) atomicAdd(&g_rank2_stamps[i], now_ - stamp_); stamp_ = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define R2_STAMP_INIT unsigned long long stamp_ = __builtin_amdgcn_s_memtime()
#else
#define R2_STAMP(i) do { } while (0)
#define R2_STAMP_INIT do { } while (0)
#endif

#ifndef WV_R2_WINBINS
#define WV_R2_WINBINS 32
#endif
#ifndef WV_R2_W25
#define WV_R2_W25 5
#endif
// Timing ablations (tools/build_variant.sh ... -DWV_R2_ABL=n; results are wrong by construction): 1 = the list is not stored,
// 2 = nothing is placed either, 3 = nothing is counted either (distance pass + scans of an empty table), 4 = distance pass only,
// 5 = full kernel but every placement store goes to the lane's own trash slot (no scatter, no bank conflicts)
#ifndef WV_R2_ABL
#define WV_R2_ABL 0
#endif
constexpr int kWinBins = WV_R2_WINBINS;      // distance bins per window
constexpr int kWinRows = kWinBins + 1;       // + the dummy row
constexpr int kMaxBins2 = 130;               // nbits <= 128

template <int WORDS>
struct QCode {
    uint64_t w[WORDS];
};

// ---- synchronisation of the TPQ threads that share a query
template <int TPQ>
__device__ __forceinline__ void group_sync()
{
    if constexpr (TPQ == 64) {
        // one wave: LDS operations of a wave complete in order; only the compiler has to be held back
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

struct Rank2Lds {
    uint32_t *table;     // [kWinRows][TPQ/2] dwords = u16 cell per (row, thread)
    uint16_t *stage;     // [k + TPQ]: ranked item numbers, then one trash slot per lane
    uint32_t *gbase;     // [kMaxBins2 + 1]: gbase[b] = rows with distance < b (filled as windows complete)
    uint32_t *tot;       // [kWinBins]
    uint32_t *misc;      // [4]: group minimum etc.
};

// average precision instead of (or beside) the list: all pointers NULL = off
struct Rank2Ap {
    const uint32_t *cls;     // class-major label bit matrix [64 * lwords][ceil(N / 32)] (rank2_labels_prepare)
    const uint64_t *qlab;    // [Q][lwords] label words of every query
    int lwords;              // 1 or 2 (up to 128 classes)
    float *ap;               // [Q]
    int32_t *nrel;           // [Q] relevant entries among the k (or NULL)
    uint64_t *relbits;       // [Q][ceil(k / 64)] instead of ap: the relevance string of the list (sharded mAP)
    int64_t relbits_ld;      // row pitch of relbits in uint64 (0 = ceil(k / 64))
    int64_t cum_ld;          // row pitch of the histograms in uint32 (0 = nbits + 2): relbits and cum may share one wire buffer
};

// words of the relevance bitmap: one bit per database row
__host__ __device__ inline int rank2_bitmap_words(int64_t N) { return (int)((N + 31) / 32); }

template <int TPQ>
__host__ __device__ inline size_t rank2_lds_bytes_per_query(int k, int bm_words = 0)
{
    size_t b = (size_t)kWinRows * (TPQ / 2) * 4;                 // table
    b += ((size_t)(k + TPQ) * 2 + 15) / 16 * 16;                 // stage
    b += (size_t)(kMaxBins2 + 1 + kWinBins + 4 + 3) / 4 * 4 * 4; // gbase, tot, misc
    b += (size_t)bm_words * 4;                                   // relevance bitmap
    b = (b + 15) / 16 * 16;
    const size_t hist = (size_t)(kMaxBins2 + 1) * 17 * 4;        // histogram-only mode: [bins + 1][16] dwords + totals
    return b > hist ? b : (hist + 15) / 16 * 16;
}

// One pass over the database image: distances of QB queries -> bytes in registers (dc[qq][i/4] byte i%4 = item i of this
// thread; 255 = no item), and a lower bound of this thread's smallest distance per query.
// Image (rank2_prepare): 16 bytes per (row, thread): two consecutive 64-bit codes of the thread, or one 128-bit code.
// Nothing in the item loop asks whether an item exists (a compare + select per item was 2 of its 7 VALU instructions):
// slots beyond the thread's C items and the padding items behind row N - 1 (zero codes in the image) get a distance like
// every other, and are overwritten with 255 afterwards -- one OR per cache WORD with a mask that is uniform (slots >= C)
// and, in the one or few threads behind the last row, one divergent fix-up.  dmin may therefore include a padding item's
// distance popcount(q): a bound BELOW the true minimum only opens the first window earlier (bins without rows), which
// is exact.
template <int WORDS, int TPQ, int NC, int QB>
__device__ __forceinline__ void rank2_distances(const uint4 *__restrict__ img, const QCode<WORDS> (&qc)[QB], int64_t N, int C,
                                                int t, uint32_t (&dc)[QB][NC], uint32_t (&dmin)[QB])
{
    constexpr int UNR = WORDS == 1 ? 16 : 8;                     // items per batch: eight 16-byte loads in flight per lane (a
    constexpr int LPB = WORDS == 1 ? UNR / 2 : UNR;              // workgroup's time is a chain of load -> popcount rounds)
    // (Round 3: a rolling ring -- consume a row, refill its slot with the row 8 further on, so that eight loads stay in flight
    // for the whole pass instead of draining to zero after every batch -- was built.  Any branch between a refill and its use
    // lets the compiler sink the load into the branch or wait vmcnt(0) at the join; the branch-free form, pinned with
    // sched_barrier, ran the distance pass in 55 us against 19.5 us for these batches (ring of 12: 57 us): eight loads per
    // round trip either way, and the batches issue theirs back to back.  What would help is a deeper ring, i.e. registers.)
    const int first = t * C;
    const int nvalid = min(C, max(0, (int)N - first));           // N < 65536, first <= 256 * 128: plain ints
    const int rows = WORDS == 1 ? (C + 1) / 2 : C;               // image rows
    const uint32_t toff = (uint32_t)t * 16u;
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) {
        dmin[qq] = 255;
#pragma unroll
        for (int i = 0; i < NC; ++i) dc[qq][i] = 0xffffffffu;
    }
#pragma unroll
    for (int bi = 0; bi < (NC * 4 + UNR - 1) / UNR; ++bi) {
        if (bi * UNR < C) {                                      // uniform
            uint4 raw[LPB];
#pragma unroll
            for (int u = 0; u < LPB; ++u)                        // uniform row base (SGPRs) + the lane's 32-bit byte offset
                raw[u] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(img) +
#ifdef WV_R2_SAMEROW   /* timing probe: every load hits the same 4 KB of the image (L1-resident) */
                                                          (size_t)min((bi * LPB + u) & 1, rows - 1) * (TPQ * 16) + toff);
#else
                                                          (size_t)min(bi * LPB + u, rows - 1) * (TPQ * 16) + toff);
#endif
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) {
#pragma unroll
                for (int u4 = 0; u4 < UNR / 4; ++u4) {
                    if (bi * (UNR / 4) + u4 >= NC) continue;     // NC odd: the last batch has one cache word
                    uint32_t word = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int u = 4 * u4 + j;
                        uint32_t d;
                        if constexpr (WORDS == 1) {
                            const uint4 v = raw[u >> 1];
                            const uint32_t lo32 = (u & 1) ? v.z : v.x, hi32 = (u & 1) ? v.w : v.y;
                            d = (uint32_t)__popc(lo32 ^ (uint32_t)qc[qq].w[0]) + (uint32_t)__popc(hi32 ^ (uint32_t)(qc[qq].w[0] >> 32));
                        } else {
                            const uint4 v = raw[u];
                            d = (uint32_t)__popc(v.x ^ (uint32_t)qc[qq].w[0]) + (uint32_t)__popc(v.y ^ (uint32_t)(qc[qq].w[0] >> 32)) +
                                (uint32_t)__popc(v.z ^ (uint32_t)qc[qq].w[1]) + (uint32_t)__popc(v.w ^ (uint32_t)(qc[qq].w[1] >> 32));
                        }
                        dmin[qq] = min(dmin[qq], d);             // slots beyond C repeat the thread's last row: real distances
                        word |= d << (8 * j);
                    }
                    dc[qq][bi * (UNR / 4) + u4] = word;
                }
            }
        }
    }
    // ---- slots that hold no item -> 255.  Uniform part: slots >= C of every thread (scalar masks)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int keep = min(4, max(0, C - 4 * i));              // item slots of word i below C
        const uint32_t fill = keep >= 4 ? 0u : (0xffffffffu << (8 * keep));
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) dc[qq][i] |= fill;
    }
    // ... and the threads behind the database's last row (nvalid < C: the last thread with rows and every one after it)
    if (nvalid < C) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int keep = min(4, max(0, nvalid - 4 * i));
            const uint32_t fill = keep >= 4 ? 0u : (0xffffffffu << (8 * keep));
#pragma unroll
            for (int qq = 0; qq < QB; ++qq) dc[qq][i] |= fill;
        }
    }
}

// Histogram only (first step of the sharded search): ONE count pass over all bins.  No per-thread cells are needed when
// nothing is placed, so 8 threads share a cell (cell = (bin, t & 31), 16-bit): nbins + 1 rows of 64 bytes.
template <int TPQ, int NC>
__device__ __forceinline__ void rank2_hist_only(const uint32_t (&dc)[NC], int64_t N, int C, int nbins,
                                                uint32_t *__restrict__ cum_out, uint8_t *lds_raw, int t)
{
    uint32_t *table = reinterpret_cast<uint32_t *>(lds_raw);     // [nbins + 1][16] dwords (the window table's space: 8.3 KB of 16.9)
    uint32_t *tot = table + (kMaxBins2 + 1) * 16;                // [kMaxBins2 + 1]
    for (int i = t; i < (nbins + 1) * 16; i += TPQ) table[i] = 0;
    group_sync<TPQ>();
    const uint32_t c = (uint32_t)(t & 31);
    const uint32_t cell_addr = (c >> 1) * 4u, cell_inc = 1u << (16 * (c & 1));
    char *tbl = reinterpret_cast<char *>(table);
#pragma unroll
    for (int bw = 0; bw < NC; bw += 2) {
        if (bw * 4 < C) {                                         // uniform
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (bw + (j >> 2) < NC) {
                    const uint32_t d = (dc[bw + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                    const uint32_t b = min(d, (uint32_t)nbins);   // 255 (no item) -> the dummy row
                    __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(tbl + b * 64 + cell_addr), cell_inc,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    group_sync<TPQ>();
    for (int b = t; b < nbins; b += TPQ) {                        // thread b sums row b
        const uint4 *row = reinterpret_cast<const uint4 *>(table + b * 16);
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 v = row[i];
            s += (v.x & 0xffffu) + (v.x >> 16) + (v.y & 0xffffu) + (v.y >> 16) + (v.z & 0xffffu) + (v.z >> 16) +
                 (v.w & 0xffffu) + (v.w >> 16);
        }
        tot[b] = s;
    }
    group_sync<TPQ>();
    if (t < 64) {                                                 // exclusive scan over the bins: 3 bins per lane >= 130
        const int b0 = 3 * t;
        const uint32_t t0 = b0 < nbins ? tot[b0] : 0u, t1 = b0 + 1 < nbins ? tot[b0 + 1] : 0u, t2 = b0 + 2 < nbins ? tot[b0 + 2] : 0u;
        const uint32_t incl = wave_incl_scan_u32(t0 + t1 + t2), excl = incl - (t0 + t1 + t2);
        if (b0 <= nbins) cum_out[b0] = excl;
        if (b0 + 1 <= nbins) cum_out[b0 + 1] = excl + t0;
        if (b0 + 2 <= nbins) cum_out[b0 + 2] = excl + t0 + t1;
    }
    group_sync<TPQ>();
    (void)N;
}

// Relevance bitmap of one query in LDS: bit (row) = the row's label word shares a bit with the query's
// (label_comparison_fn, accuracy_calculator.py:31-37, on multi-hot words).  The labels come as a class-major bit matrix
// cls[64][nw] (bit r of row c = row r carries class c: rank2_labels_prepare), so the bitmap is the OR of the matrix rows
// of the query's classes -- a few 3 KB rows instead of all N label words (a pass over N x 8 bytes per query, the size of
// the code image, cost as much as the distance pass: the texture path moves 64 bytes per clock and CU).
template <int TPQ>
__device__ __forceinline__ void rank2_relevance_bitmap(const uint32_t *__restrict__ cls, uint64_t ql, uint64_t ql_hi, int nw, int t,
                                                       uint32_t *bitmap)
{
    // A thread owns words t, t + TPQ, ... (at most 4: N <= 32768).  Per round four classes x four words = 16 loads are
    // issued before the first is used: an L2 round trip costs ~4 k cycles under this kernel's load, a word-by-word,
    // class-by-class chain of them was a third of the workgroup's time.
    constexpr int WPT = 4;
    uint32_t acc[WPT] = {0u, 0u, 0u, 0u};
    int wi[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) wi[i] = min(t + i * TPQ, nw - 1);
    uint64_t m = ql;                                             // uniform: the loop runs on the scalar unit
    int cbase = 0;                                               // classes 0-63, then (two label words) 64-127
    if (!m) { m = ql_hi; ql_hi = 0; cbase = 64; }
    while (m) {
        const int c0 = cbase + __builtin_ctzll(m);
        m &= m - 1;
        const int c1 = m ? cbase + __builtin_ctzll(m) : c0;
        m &= m - 1;
        const int c2 = m ? cbase + __builtin_ctzll(m) : c0;
        m &= m - 1;
        const int c3 = m ? cbase + __builtin_ctzll(m) : c0;
        m &= m - 1;
        uint32_t v[4][WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            v[0][i] = cls[(size_t)c0 * nw + wi[i]];
            v[1][i] = cls[(size_t)c1 * nw + wi[i]];
            v[2][i] = cls[(size_t)c2 * nw + wi[i]];
            v[3][i] = cls[(size_t)c3 * nw + wi[i]];
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) acc[i] |= (v[0][i] | v[1][i]) | (v[2][i] | v[3][i]);
        if (!m && ql_hi) { m = ql_hi; ql_hi = 0; cbase = 64; }   // second label word
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
        if (t + i * TPQ < nw) bitmap[t + i * TPQ] = acc[i];
    for (int w = t + WPT * TPQ; w < nw; w += TPQ) bitmap[w] = 0;   // not reached for N <= 32768 rows and TPQ = 256
    // the caller's next group barrier (there are several before the list is walked) makes the bitmap visible
}

// AP of the list in LDS (k_map_at_k's arithmetic and order: position p = round * TPQ + t, the j-th hit adds the fp32
// quotient j / (p + 1) to a double, waves summed in index order).  scratch: free LDS, (32 * NW + 2 * NW + 2) dwords.
// relbits_out (or NULL): the relevance string itself, bit p of the uint64 array = relevance of list position p -- what a
// shard contributes to the sharded mAP (wv_hamming_shard_relbits); no AP is computed then.
template <int TPQ>
__device__ __forceinline__ void rank2_ap(const uint16_t *stage, const uint32_t *bitmap, uint32_t *scratch, int k, int t,
                                         float *__restrict__ ap_out, int32_t *__restrict__ nrel_out,
                                         uint64_t *__restrict__ relbits_out)
{
    constexpr int NW = TPQ / 64;
    const int lane = t & 63, wv = t >> 6;
    const int R = (k + TPQ - 1) / TPQ;                           // rounds; walked in chunks of kApRounds
    uint32_t *cnt = scratch;                                     // [chunk rounds][NW] hits of a wave in a round
    double *wsum = reinterpret_cast<double *>(scratch + kApRounds * NW + (kApRounds * NW & 1));
    // Eight rounds at a time, every LDS read of a batch issued before the first is used: with other workgroups' atomics
    // queued at the LDS unit a read takes ~1k cycles, and a chain of dependent ones (list entry -> bitmap word, round
    // after round) would pay that 2 R times.
    constexpr int CH = 8;
    ApState st;
    for (int c0 = 0; c0 < R; c0 += kApRounds) {                  // one chunk for k <= 32 * TPQ (mAP@5000); more for mAP@ALL
        const int Rc = min(kApRounds, R - c0);
        uint32_t relbits = 0;
        for (int r0 = c0; r0 < c0 + Rc; r0 += CH) {
            uint32_t it[CH], wd[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) it[u] = stage[min((r0 + u) * TPQ + t, k - 1)];
#pragma unroll
            for (int u = 0; u < CH; ++u) wd[u] = bitmap[it[u] >> 5];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int r = r0 + u;
                const bool rel = r < c0 + Rc && r * TPQ + t < k && ((wd[u] >> (it[u] & 31)) & 1u);
                relbits |= (rel ? 1u : 0u) << ((r - c0) & 31);
                const uint64_t m = __ballot(rel);
                if (lane == 0 && r < c0 + Rc) {
                    if (!relbits_out) cnt[(r - c0) * NW + wv] = (uint32_t)__popcll(m);
                    else if ((r * NW + wv) * 64 < k) relbits_out[r * NW + wv] = m;
                }
            }
        }
        if (relbits_out) continue;                               // a shard of the sharded mAP: the string is all it sends
        group_sync<TPQ>();                                       // the chunk's hit counts are published
        ap_accum<TPQ>(relbits, cnt, Rc, c0, t, st);
        group_sync<TPQ>();                                       // ... and read: the next chunk may overwrite them
    }
    if (relbits_out) return;
    ap_final<TPQ>(st, wsum, t, ap_out, nrel_out, [] { group_sync<TPQ>(); });
}
template <int TPQ>
__device__ __forceinline__ Rank2Lds rank2_lds(uint8_t *lds_raw, int k)
{
    Rank2Lds L;
    L.table = reinterpret_cast<uint32_t *>(lds_raw);
    uint8_t *p = lds_raw + (size_t)kWinRows * TPQ * 2;
    L.stage = reinterpret_cast<uint16_t *>(p);
    p += ((size_t)(k + TPQ) * 2 + 15) / 16 * 16;
    L.gbase = reinterpret_cast<uint32_t *>(p);
    L.tot = L.gbase + kMaxBins2 + 1;
    L.misc = L.tot + kWinBins;
    return L;
}

template <int TPQ>
__device__ __forceinline__ void rank2_zero_table(uint8_t *lds_raw, int t)
{
    uint4 *t4 = reinterpret_cast<uint4 *>(lds_raw);
    constexpr int n4 = kWinRows * TPQ * 2 / 16;
    for (int i = t; i < n4; i += TPQ) t4[i] = make_uint4(0, 0, 0, 0);
}

// The ranked list (staged in LDS by rank2_rank) leaves for global memory.
template <int TPQ>
__device__ __forceinline__ void rank2_copy_list(uint8_t *lds_raw, int k, int64_t idx_offset, int32_t *__restrict__ idx_out,
                                                uint16_t *__restrict__ rows16_out, int t)
{
    const Rank2Lds L = rank2_lds<TPQ>(lds_raw, k);
    // ---- the ranked list leaves with 16-byte stores
    if (rows16_out) {
        if ((k & 7) == 0 && (reinterpret_cast<uintptr_t>(rows16_out) & 15) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(L.stage);
            uint4 *o4 = reinterpret_cast<uint4 *>(rows16_out);
            for (int i = t; i < k / 8; i += TPQ) o4[i] = s4[i];
        } else {
            for (int i = t; i < k; i += TPQ) rows16_out[i] = L.stage[i];
        }
    } else if (idx_out) {
        const bool vec = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(idx_out) & 15) == 0;
        const int32_t off = (int32_t)idx_offset;
        if (vec) {
            const uint2 *s2 = reinterpret_cast<const uint2 *>(L.stage);
            int4 *o4 = reinterpret_cast<int4 *>(idx_out);
            for (int i = t; i < k / 4; i += TPQ) {
                const uint2 v = s2[i];
                o4[i] = make_int4((int32_t)(v.x & 0xffffu) + off, (int32_t)(v.x >> 16) + off,
                                  (int32_t)(v.y & 0xffffu) + off, (int32_t)(v.y >> 16) + off);
            }
        } else {
            for (int i = t; i < k; i += TPQ) idx_out[i] = (int32_t)L.stage[i] + off;
        }
    }
}

// The distance row of the ranked list, regenerated from the bin boundaries: dist[p] = b with gbase[b] <= p < gbase[b+1]; 16
// positions per thread.  Callers that only need the list -- mAP -- leave dist_out NULL.  A search-free form (boundaries in
// registers, wave-uniform v_readlane walk with packed byte adds) was built and measured slower.
template <int TPQ>
__device__ __forceinline__ void rank2_dist_row(const Rank2Lds &L, int nbins, int k, uint8_t *__restrict__ dist_out, int t)
{
    {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(dist_out);
        for (int p0 = t * 16; p0 < k; p0 += TPQ * 16) {
            int a = 0, z = nbins;                                 // invariant: gbase[a] <= p0 < gbase[z]
            while (z - a > 1) {
                const int mid = (a + z) >> 1;
                if (L.gbase[mid] <= (uint32_t)p0) a = mid;
                else z = mid;
            }
            int bin = a;
            uint32_t next = L.gbase[bin + 1];
            uint32_t w4[4];
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t pp = (uint32_t)(p0 + 4 * j4 + j);
                    while (pp >= next && bin + 1 < nbins) { ++bin; next = L.gbase[bin + 1]; }
                    word |= (uint32_t)bin << (8 * j);
                }
                w4[j4] = word;
            }
            if (((a0 + p0) & 15) == 0 && p0 + 16 <= k) {
                *reinterpret_cast<uint4 *>(dist_out + p0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            } else if (((a0 + p0) & 3) == 0) {
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    if (p0 + 4 * j4 + 4 <= k) {
                        *reinterpret_cast<uint32_t *>(dist_out + p0 + 4 * j4) = w4[j4];
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (p0 + 4 * j4 + j < k) dist_out[p0 + 4 * j4 + j] = (uint8_t)(w4[j4] >> (8 * j));
                    }
                }
            } else {
                for (int j = 0; j < 16; ++j)
                    if (p0 + j < k) dist_out[p0 + j] = (uint8_t)(w4[j >> 2] >> (8 * (j & 3)));
            }
        }
    }
}

// Ranks ONE query from its cached distances into the staged list in LDS (rank2_copy_list moves it to global memory).
// Expects the count table zeroed (rank2_zero_table + a group barrier before the first LDS add: the caller's -- the
// zeroing overlaps the distance pass).  Ends with a group barrier.
// NC = distance-cache words (4 items each): items per thread C <= 4 * NC
template <int TPQ, int NC, bool AP = false>
__device__ __forceinline__ void rank2_rank(const uint32_t (&dc)[NC], uint32_t dmin, int64_t N, int C, int nbins, int k,
                                           uint32_t *__restrict__ cum_out, uint8_t *__restrict__ dist_out, uint8_t *lds_raw, int t,
                                           const uint32_t *__restrict__ cls = nullptr, uint64_t qlabel = 0, uint64_t qlabel_hi = 0,
                                           float *__restrict__ ap_out = nullptr, int32_t *__restrict__ nrel_out = nullptr,
                                           uint64_t *__restrict__ relbits_out = nullptr)
{
    constexpr int ROWB = TPQ * 2;                                // bytes per table row
    const Rank2Lds L = rank2_lds<TPQ>(lds_raw, k);
    uint32_t *bitmap = L.misc + 4;                               // relevance of every item (AP only)
    R2_STAMP_INIT;
    if constexpr (AP) rank2_relevance_bitmap<TPQ>(cls, qlabel, qlabel_hi, rank2_bitmap_words(N), t, bitmap);
    R2_STAMP(6);
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform
    constexpr int NW = TPQ / 64;                                 // waves per query
    const int first = t * C;
    // smallest distance of the query = first bin of the first window
    dmin = wave_min_u32(dmin);
    if constexpr (NW > 1) {
        if (lane == 0) L.misc[wv] = dmin;
        __syncthreads();
        dmin = min(min(L.misc[0], L.misc[1]), min(L.misc[2], L.misc[3]));
    }
    // bins below the first window are empty
    for (int b = t; b <= nbins; b += TPQ) L.gbase[b] = b <= (int)dmin ? 0u : 0xffffffffu;
    int lo = min((int)dmin, nbins - 1);                          // dmin == 255 cannot happen (N >= 1)
    uint32_t placed = 0;                                         // rows with distance < lo
    // Cell of thread t inside a table row: dword (t>>6)*32 + (t&31), half (t>>5)&1 -- lanes l and l+32 of a wave share a
    // dword.  A wave's LDS instruction is served in two groups of 32 lanes; this way each group touches 32 different
    // banks whatever the bins are (pairing lanes 2j, 2j+1 instead made every atomic a 2-way conflict).
    // A cell counts in BYTES of the 16-bit list (2 per item): what a placement add returns is the item's byte offset in
    // the staged list, no shift between the atomic and the store.  Cells start at 2 * min(rank, k) (2k for the dummy row)
    // and take at most C <= 128 items: 2 * (k + 128) < 65536 (host check).
    const uint32_t cell_addr = (uint32_t)((t >> 6) * 32 + (t & 31)) * 4u;
    const uint32_t cell_shift = 16u * ((t >> 5) & 1);
    const uint32_t cell_inc = 2u << cell_shift;
    char *tbl = reinterpret_cast<char *>(L.table);
    char *stage_b = reinterpret_cast<char *>(L.stage);
    const uint32_t trash = 2u * (uint32_t)(k + t);
    const uint32_t k2 = 2u * (uint32_t)k;

    R2_STAMP(0);
    for (;;) {
        const bool place = placed < (uint32_t)k;                 // uniform: false = count-only pass (cum requested)
        // ---- count: one LDS add per item, into the row of its bin or into the dummy row (items below the window --
        // placed by an earlier one -- wrap to huge values and clamp to the dummy row too).  Batches of 8 items (two cache
        // words): one uniform branch per batch; the adds return nothing, so they simply queue up.
        // (Addressing the row by the ABSOLUTE distance, clamped from above only -- 2 VALU instructions per item instead of
        // 3 -- was built in round 3: it needs the cache bytes below a later window rewritten to 255, and that rarely taken
        // path cost registers (28-64 bytes of scratch per thread) for no measurable gain: the item loops are not what bounds
        // the launch)
        char *row0 = tbl + cell_addr;
#pragma unroll
        for (int bw = 0; bw < NC; bw += 2) {
            if (bw * 4 < C && (WV_R2_ABL < 3 || WV_R2_ABL == 5 || k < 0)) {         // uniform; items >= C of the batch hold 255 -> dummy row
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (bw + (j >> 2) < NC) {
                        const uint32_t d = (dc[bw + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                        const uint32_t b = min(d - (uint32_t)lo, (uint32_t)kWinBins);
                        __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(row0 + b * ROWB), cell_inc,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        group_sync<TPQ>();
        R2_STAMP(1);
        // ---- per-bin totals and the threads' exclusive prefixes inside each bin
        uint32_t win_total, bin_base;
        if constexpr (TPQ == 64) {
            for (int b = 0; b < kWinBins; ++b) {
                uint32_t s2 = (L.table[b * (TPQ / 2) + (lane & 31)] >> cell_shift) & 0xffffu;
                s2 = wave_sum_u32(s2) >> 1;                       // the cells count bytes of the list
                if (lane == 0) L.tot[b] = s2;
            }
            group_sync<TPQ>();
            const uint32_t my_tot = lane < kWinBins ? L.tot[lane] : 0u;
            const uint32_t incl = wave_incl_scan_u32(my_tot);
            bin_base = placed + incl - my_tot;                    // lane b: rows with distance < lo + b
            win_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (lane < kWinBins && lo + lane + 1 <= nbins) L.gbase[lo + lane + 1] = placed + incl;
            if (place) {
                for (int b = 0; b < kWinBins; ++b) {
                    const uint32_t base_b = (uint32_t)__builtin_amdgcn_readlane((int)bin_base, b);
                    const uint32_t c = (L.table[b * (TPQ / 2) + (lane & 31)] >> cell_shift) & 0xffffu;
                    const uint32_t excl = wave_incl_scan_u32(c) - c + 2u * base_b;
                    reinterpret_cast<uint16_t *>(L.table + b * (TPQ / 2))[2 * (lane & 31) + (lane >> 5)] = (uint16_t)min(excl, k2);
                }
            }
        } else {
            // Two passes over the wave's bins (wv, wv + NW, ...): totals, then the threads' exclusive prefixes inside each bin.
            // The phase is a chain of latencies (LDS read -> wave scan -> LDS write, bin after bin: 5 k of the 19 k cycles
            // a query takes alone on a CU).  Issuing the reads of 2 / 4 / 8 bins before the first scan and running their scans
            // side by side was built and measured (round 3): no gain at c1 (44.5 us bin by bin, 44.6 / 44.4 / 50.6 us --
            // from 4 bins on the registers spill); keeping the first pass's prefixes in registers, one scan per bin
            // instead of two, spills too.  With five workgroups per CU other queries fill the gaps: the launch is bound by
            // the SUM of the CU's vector-memory, VALU and LDS work, not by this chain.
            for (int b = wv; b < kWinBins; b += NW) {
                const uint2 v = *reinterpret_cast<const uint2 *>(L.table + b * (TPQ / 2) + 2 * lane);   // any 4 cells
                uint32_t s2 = (v.x & 0xffffu) + (v.x >> 16) + (v.y & 0xffffu) + (v.y >> 16);
                s2 = wave_sum_u32(s2) >> 1;                       // the cells count bytes of the list
                if (lane == 0) L.tot[b] = s2;
            }
            group_sync<TPQ>();
            const uint32_t my_tot = lane < kWinBins ? L.tot[lane] : 0u;
            const uint32_t incl = wave_incl_scan_u32(my_tot);
            bin_base = placed + incl - my_tot;                    // lane b: rows with distance < lo + b
            win_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (wv == 0 && lane < kWinBins && lo + lane + 1 <= nbins) L.gbase[lo + lane + 1] = placed + incl;
            if (place) {
                // lane L scans threads 4L .. 4L+3: four consecutive dwords of the row, the same half of each
                const int t0 = 4 * lane, d0 = (t0 >> 6) * 32 + (t0 & 31), sh = 16 * ((t0 >> 5) & 1);
                for (int b = wv; b < kWinBins; b += NW) {
                    const uint32_t base_b = (uint32_t)__builtin_amdgcn_readlane((int)bin_base, b);
                    const uint4 v = *reinterpret_cast<const uint4 *>(L.table + b * (TPQ / 2) + d0);
                    const uint32_t c0 = (v.x >> sh) & 0xffffu, c1 = (v.y >> sh) & 0xffffu, c2 = (v.z >> sh) & 0xffffu,
                                   c3 = (v.w >> sh) & 0xffffu;
                    const uint32_t s4 = c0 + c1 + c2 + c3;
                    // cells restart at 2 * min(rank of the cell's first item, k)
                    const uint32_t e0 = wave_incl_scan_u32(s4) - s4 + 2u * base_b;
                    const uint32_t e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2;
                    // the other half of these dwords belongs to another lane: 16-bit stores
                    uint16_t *h = reinterpret_cast<uint16_t *>(L.table + b * (TPQ / 2) + d0) + (sh >> 4);
                    h[0] = (uint16_t)min(e0, k2); h[2] = (uint16_t)min(e1, k2); h[4] = (uint16_t)min(e2, k2); h[6] = (uint16_t)min(e3, k2);
                }
            }
        }
        if (place) {
            // dummy row: every cell starts at 2k, so whatever it returns is >= 2k (trash); 2 (k + 128) < 65536 (host check)
            if (wv == NW - 1) {
                const uint32_t kk = k2 * 0x10001u;
                for (int i = lane; i < TPQ / 2; i += 64) L.table[kWinBins * (TPQ / 2) + i] = kk;
            }
            group_sync<TPQ>();
            R2_STAMP(2);
            // ---- placement: returning LDS add = this item's byte offset in the list, item number into the LDS list (or
            // the trash slot).  Sixteen returning adds are in flight before the first result is used (the round trip is the
            // cost here).
#pragma unroll
            for (int bw = 0; bw < NC; bw += 4) {
                if (bw * 4 < C && (WV_R2_ABL < 2 || WV_R2_ABL == 5 || k < 0)) {     // uniform
                    uint32_t old[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        old[j] = 0;
                        if (bw + (j >> 2) < NC) {
                            const uint32_t d = (dc[bw + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                            const uint32_t b = min(d - (uint32_t)lo, (uint32_t)kWinBins);
                            old[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(row0 + b * ROWB), cell_inc,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        if (bw + (j >> 2) < NC) {
                            const uint32_t pos2 = (old[j] >> cell_shift) & 0xffffu;      // byte offset in the list
                            *reinterpret_cast<uint16_t *>(stage_b + min(WV_R2_ABL == 5 ? (pos2 | 0x1ffffu) : pos2, trash)) =
                                (uint16_t)(first + bw * 4 + j);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        R2_STAMP(3);
        placed += win_total;
        lo += kWinBins;
        // uniform exit: the list is complete and nobody asked for the full histogram, or no bins are left
        if (lo >= nbins || (placed >= (uint32_t)k && !cum_out)) break;
        group_sync<TPQ>();                                        // every add of this window has returned
        rank2_zero_table<TPQ>(lds_raw, t);
        group_sync<TPQ>();
    }
    group_sync<TPQ>();
    // bins beyond the last window processed hold every row (the loop only stops early once placed >= k)
    // ---- cumulative histogram of all rows (cum[b] = rows with distance < b), for the sharded search
    if (cum_out)
        for (int b = t; b <= nbins; b += TPQ) cum_out[b] = min(L.gbase[b], (uint32_t)N);
    if (dist_out) rank2_dist_row<TPQ>(L, nbins, k, dist_out, t);
    R2_STAMP(4);
    if constexpr (AP) {
        rank2_ap<TPQ>(L.stage, bitmap, L.table, k, t, ap_out, nrel_out, relbits_out);   // the count table is free by now
        R2_STAMP(7);
        group_sync<TPQ>();                                        // the table (AP scratch) and the bitmap are free again
    }
}

// Image for TPQ threads per query, 16 bytes per (row, thread).  64-bit codes: row r2 of thread t = its items 2*r2 and
// 2*r2 + 1 (item = t*C + i, C = ceil(N / TPQ)); 128-bit codes: row r = item r.  Items beyond the thread's range or beyond
// N are zero (the kernel masks them by index, never by value).
template <int WORDS>
__global__ __launch_bounds__(256) void k_rank2_image(const uint64_t *__restrict__ db, uint4 *__restrict__ img, int64_t N,
                                                     int C, int tpq)
{
    const int rows = WORDS == 1 ? (C + 1) / 2 : C;
    const int64_t total = (int64_t)rows * tpq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / tpq;
        const int t = (int)(i - r * tpq);
        uint64_t a = 0, b = 0;
        if constexpr (WORDS == 1) {
            const int64_t i0 = 2 * r, i1 = 2 * r + 1, it0 = (int64_t)t * C + i0, it1 = (int64_t)t * C + i1;
            if (i0 < C && it0 < N) a = db[it0];
            if (i1 < C && it1 < N) b = db[it1];
        } else {
            const int64_t it = (int64_t)t * C + r;
            if (it < N) { a = db[it * 2]; b = db[it * 2 + 1]; }
        }
        img[i] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
}

// Class-major label bit matrix: cls[c][w] bit j = row 32 w + j carries class c (bit c % 64 of its label word c / 64).
__global__ __launch_bounds__(256) void k_rank2_label_matrix(const uint64_t *__restrict__ dblab, uint32_t *__restrict__ cls, int64_t N,
                                                            int nw, int lwords)
{
    const int ncls = 64 * lwords;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread per (word, class), the class fastest
    if (i >= (int64_t)nw * ncls) return;
    const int c = (int)(i % ncls);
    const int64_t w = i / ncls;
    uint32_t word = 0;
    for (int j = 0; j < 32; ++j) {
        const int64_t r = w * 32 + j;
        if (r < N) word |= (uint32_t)((dblab[r * lwords + (c >> 6)] >> (c & 63)) & 1ull) << j;
    }
    cls[(size_t)c * nw + w] = word;
}

size_t rank2_labels_bytes(int64_t N, int lwords) { return (size_t)64 * lwords * rank2_bitmap_words(N) * sizeof(uint32_t); }

int rank2_labels_prepare(const uint64_t *dblab, void *cls, int64_t N, int lwords, hipStream_t st)
{
    const int nw = rank2_bitmap_words(N);
    hipLaunchKernelGGL(k_rank2_label_matrix, dim3((unsigned)ceil_div((int64_t)nw * 64 * lwords, 256)), dim3(256), 0, st, dblab,
                       (uint32_t *)cls, N, nw, lwords);
    WV_CHECK_LAUNCH("k_rank2_label_matrix");
    return WV_OK;
}

// minimum waves per SIMD the register allocation has to leave room for (the LDS footprint admits at least as many)
constexpr int rank2_min_waves(int nc, int qb) { return nc * qb <= 8 ? 7 : (nc * qb <= 16 ? 6 : (nc * qb <= 25 ? WV_R2_W25 : (nc * qb <= 50 ? 4 : 3))); }

// AP = true: the instantiation behind wv_hamming_map_at_k (its own kernels: the plain ranking keeps its registers)
//
// One query per group of TPQ threads.  A persistent form -- the grid sized to what the chip holds at once, a group looping
// over queries, the list of query i leaving LDS only after the distance pass of query i + 1 so that its stores drain
// behind the LDS-only phases -- was built and measured (round 3, c1): 1280 resident workgroups 41.3 us, 2048 one-query
// workgroups 41.0 us, 1024 workgroups x 2 queries 47.5 us, 256 x 8 96 us.  The hardware's dispatch already is that queue;
// a query's time is a chain of latencies (12 us for a workgroup alone on its CU) that co-resident workgroups stretch to
// ~20 us at five per CU, whatever order the phases are issued in.
template <int WORDS, int TPQ, int NC, int QB, bool AP>
__global__ __launch_bounds__(256, rank2_min_waves(NC, QB)) void k_rank_window(const uint64_t *__restrict__ q, const uint4 *__restrict__ img,
                                                     int32_t *__restrict__ idx, uint16_t *__restrict__ rows16,
                                                     uint8_t *__restrict__ dist, int Q, int64_t N, int C, int nbins, int k,
                                                     int64_t idx_offset, uint32_t *__restrict__ cum, int lds_per_group, Rank2Ap apx)
{
    extern __shared__ uint4 lds4[];
    constexpr int GPW = 256 / TPQ;                              // query groups per workgroup
    const int g = threadIdx.x / TPQ, t = threadIdx.x % TPQ;
    const int q0 = (blockIdx.x * GPW + g) * QB;                 // first query of this group
    if (q0 >= Q) return;                                         // whole waves only (TPQ == 64): no barrier is skipped
    uint8_t *lds = reinterpret_cast<uint8_t *>(lds4) + (size_t)g * lds_per_group;
    QCode<WORDS> qc[QB];
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) {
        const int qi = min(q0 + qq, Q - 1);                      // tail group: recompute the last query, never store it twice
#pragma unroll
        for (int w = 0; w < WORDS; ++w) {
            const uint64_t v = q[(int64_t)qi * WORDS + w];
            // the query is uniform over its group: keep it in SGPRs
            qc[qq].w[w] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
        }
    }
    if (k != 0) rank2_zero_table<TPQ>(lds, t);                  // LDS stores in the shadow of the image loads
    uint32_t dc[QB][NC], dmin[QB];
    rank2_distances<WORDS, TPQ, NC, QB>(img, qc, N, C, t, dc, dmin);
#pragma unroll
    for (int qq = 0; qq < QB; ++qq) {
        const int qi = q0 + qq;
        if (qi >= Q) break;                                      // uniform over the group
        if (k == 0) {                                            // histogram only
            rank2_hist_only<TPQ, NC>(dc[qq], N, C, nbins, cum + (int64_t)qi * (nbins + 1), lds, t);
            continue;
        }
        if (qq > 0) rank2_zero_table<TPQ>(lds, t);              // (the previous query's rank ended with a group barrier)
        group_sync<TPQ>();                                       // the count table is zero
        if (WV_R2_ABL == 4 && k > 0) {                          // ablation: keep the distances alive, stop here
            uint32_t x = dmin[qq];
#pragma unroll
            for (int i = 0; i < NC; ++i) x ^= dc[qq][i];
            if (x == 0x12345678u && idx) idx[(int64_t)qi * k] = (int32_t)x;
            continue;
        }
        if constexpr (AP) {
            const uint64_t lw = apx.qlab[(int64_t)qi * apx.lwords], lw2 = apx.lwords > 1 ? apx.qlab[(int64_t)qi * apx.lwords + 1] : 0;
            const uint64_t ql = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(lw >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lw);
            const uint64_t ql2 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(lw2 >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lw2);
            rank2_rank<TPQ, NC, true>(dc[qq], dmin[qq], N, C, nbins, k,
                                      cum ? cum + (int64_t)qi * (apx.cum_ld ? apx.cum_ld : nbins + 1) : nullptr, nullptr, lds, t,
                                      apx.cls, ql, ql2, apx.ap ? apx.ap + qi : nullptr, apx.nrel ? apx.nrel + qi : nullptr,
                                      apx.relbits ? apx.relbits + (int64_t)qi * (apx.relbits_ld ? apx.relbits_ld : (k + 63) / 64)
                                                  : nullptr);
        } else {
            rank2_rank<TPQ, NC>(dc[qq], dmin[qq], N, C, nbins, k, cum ? cum + (int64_t)qi * (nbins + 1) : nullptr,
                                dist ? dist + (int64_t)qi * k : nullptr, lds, t);
            if (WV_R2_ABL == 0 || WV_R2_ABL == 5 || k < 0)
                rank2_copy_list<TPQ>(lds, k, idx_offset, idx ? idx + (int64_t)qi * k : nullptr,
                                     rows16 ? rows16 + (int64_t)qi * k : nullptr, t);
            if (qq + 1 < QB) group_sync<TPQ>();                  // the list has left the stage before the next query fills it
        }
    }
}

// ------------------------------------------------------------------------------------------ host side
// which thread count per query a shape takes: 0 = not covered by this kernel
int rank2_tpq(int Q, int64_t N, int k)
{
    const char *force = ::wv::tune("WV_TOPK_V2");                   // "0": off, "64" / "256": pin the variant (tests, tuning)
    if (force && force[0] == '0') return 0;
    if (N >= 65536 || 2 * (k + 128) >= 65536) return 0;         // list cells count bytes in 16 bits
    const bool fits256 = N <= 256 * 128 && rank2_lds_bytes_per_query<256>(k) <= 100 * 1024;
    const bool fits64 = ceil_div(N, 64) <= 64 && 4 * rank2_lds_bytes_per_query<64>(k) <= 100 * 1024;
    if (force && atoi(force) == 64 && fits64) return 64;        // a pinned variant that does not fit the shape is ignored
    if (force && atoi(force) == 256 && fits256) return 256;
    // one wave per query pays when each query has little work and there are enough queries to fill the chip
    if (fits64 && N <= 4096 && Q >= 4096) return 64;
    return fits256 ? 256 : (fits64 ? 64 : 0);
}

size_t rank2_image_bytes(int64_t N, int words, int tpq)
{
    const int64_t C = ceil_div(N, tpq), rows = words == 1 ? (C + 1) / 2 : C;
    return (size_t)rows * tpq * 16;
}

int rank2_prepare(const uint64_t *db, void *img, int64_t N, int words, int tpq, hipStream_t st)
{
    const int C = (int)ceil_div(N, tpq);
    const int64_t total = (int64_t)(words == 1 ? (C + 1) / 2 : C) * tpq;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, 256), 4096);
    if (words == 1)
        hipLaunchKernelGGL((k_rank2_image<1>), dim3(grid), dim3(256), 0, st, db, (uint4 *)img, N, C, tpq);
    else
        hipLaunchKernelGGL((k_rank2_image<2>), dim3(grid), dim3(256), 0, st, db, (uint4 *)img, N, C, tpq);
    WV_CHECK_LAUNCH("k_rank2_image");
    return WV_OK;
}

template <int WORDS, int TPQ, int NC, int QB>
static int launch_rank2_qb(const uint64_t *q, const void *img, int32_t *idx, uint16_t *rows16, uint8_t *dist, int Q, int64_t N,
                           int C, int nbins, int k, int64_t idx_offset, uint32_t *cum, const Rank2Ap &apx, hipStream_t st)
{
    constexpr int GPW = 256 / TPQ;
    const size_t per_g = rank2_lds_bytes_per_query<TPQ>(k, (apx.ap || apx.relbits) ? rank2_bitmap_words(N) : 0);
    size_t lds = per_g * GPW;
    if (const char *pad = ::wv::tune("WV_R2_PAD_LDS")) lds += (size_t)atoi(pad);   // diagnostic build: fewer workgroups per CU
    auto kern = (apx.ap || apx.relbits) ? k_rank_window<WORDS, TPQ, NC, QB, true> : k_rank_window<WORDS, TPQ, NC, QB, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "rank_window: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    const int64_t grid = ceil_div(Q, GPW * QB);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, q, (const uint4 *)img, idx, rows16, dist,
                       Q, N, C, nbins, k, idx_offset, cum, (int)per_g, apx);
    WV_CHECK_LAUNCH("k_rank_window");
    return WV_OK;
}

template <int WORDS, int TPQ, int NC, int QBMAX>
static int launch_rank2_nc(const uint64_t *q, const void *img, int32_t *idx, uint16_t *rows16, uint8_t *dist, int Q, int64_t N,
                           int C, int nbins, int k, int64_t idx_offset, uint32_t *cum, const Rank2Ap &apx, hipStream_t st)
{
    // Sharing one pass over the image between QBMAX queries of a group (ranked one after the other) was measured on
    // MI355X three times: round 2 (c1: 66 vs 50 us); round 3 with the distance pass known to be 45 % of the launch (19.6 of
    // 43.7 us with everything after it removed) and the leaner item loops, at 168 VGPRs / 3 waves per SIMD: 62-64 vs 43-45 us,
    // histogram-only launches 33 vs 22 us; and capped at 128 VGPRs / 4 waves per SIMD (8 queries resident per CU instead of 5):
    // lists 47.2-47.6 vs 44.2-45.9 us, ranking + AP 63-65 vs 59.4-59.9, histogram-only 22.3 vs 22.6 -- half the loads in the
    // same time.  What the distance pass needs is bytes in flight (latency x concurrency), not fewer loads.  The kernel keeps
    // the template parameter; only QB = 1 is instantiated.
    (void)QBMAX;
    return launch_rank2_qb<WORDS, TPQ, NC, 1>(q, img, idx, rows16, dist, Q, N, C, nbins, k, idx_offset, cum, apx, st);
}

template <int WORDS, int TPQ>
static int launch_rank2_t(const uint64_t *q, const void *img, int32_t *idx, uint16_t *rows16, uint8_t *dist, int Q, int64_t N,
                          int nbins, int k, int64_t idx_offset, uint32_t *cum, const Rank2Ap &apx, hipStream_t st)
{
    const int C = (int)ceil_div(N, TPQ);
#define WV_R2(NCW, QBM) return launch_rank2_nc<WORDS, TPQ, NCW, QBM>(q, img, idx, rows16, dist, Q, N, C, nbins, k, idx_offset, cum, apx, st)
    if (C <= 16) WV_R2(4, 8);
    if (C <= 32) WV_R2(8, 4);
    if (C <= 64) WV_R2(16, 2);
    if constexpr (TPQ == 256) {
        if (C <= 100) WV_R2(25, 2);
        WV_R2(32, 2);
    } else {
        return 1;                                                // one wave per query is for short rows only
    }
#undef WV_R2
}

// img must be the image for `tpq` threads per query (rank2_prepare)
// idx (int32 global indices) or rows16 (16-bit local row numbers) receives the list; k == 0: histogram only
int rank2_launch(const uint64_t *q, const void *img, int32_t *idx, uint16_t *rows16, uint8_t *dist, int Q, int64_t N, int nbits,
                 int k, int64_t idx_offset, uint32_t *cum, int tpq, hipStream_t st, const void *lab_img, const uint64_t *qlab,
                 float *ap, int32_t *nrel, uint64_t *relbits, int64_t relbits_ld, int64_t cum_ld, int lwords)
{
    const int nbins = nbits + 1, words = (nbits + 63) / 64;
    Rank2Ap apx{reinterpret_cast<const uint32_t *>(lab_img), qlab, lwords, ap, nrel, relbits, relbits_ld, cum_ld};
    if (ap || relbits) {
        if (!lab_img || !qlab || k < 1 || lwords < 1 || lwords > 2) return 1;
        const size_t per_g = tpq == 64 ? 4 * rank2_lds_bytes_per_query<64>(k, rank2_bitmap_words(N))
                                       : rank2_lds_bytes_per_query<256>(k, rank2_bitmap_words(N));
        if (per_g > 100 * 1024) return 1;
    }
    if (words == 1) {
        if (tpq == 64) return launch_rank2_t<1, 64>(q, img, idx, rows16, dist, Q, N, nbins, k, idx_offset, cum, apx, st);
        return launch_rank2_t<1, 256>(q, img, idx, rows16, dist, Q, N, nbins, k, idx_offset, cum, apx, st);
    }
    if (tpq == 64) return launch_rank2_t<2, 64>(q, img, idx, rows16, dist, Q, N, nbins, k, idx_offset, cum, apx, st);
    return launch_rank2_t<2, 256>(q, img, idx, rows16, dist, Q, N, nbins, k, idx_offset, cum, apx, st);
}

}  // namespace wv

#ifdef WV_RANK2_STAMPS
// diagnostic build only: read and reset the phase cycle sums (host array of 8)
extern "C" int wv_debug_rank2_stamps(unsigned long long *host8)
{
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(wv::g_rank2_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -5;
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(wv::g_rank2_stamps), zero, sizeof(zero)) != hipSuccess) return -5;
    return 0;
}
#endif

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
// Covers N < 65536 (16-bit item numbers / counters), C <= 128 items per thread (distances cached in registers as
// bytes), k small enough for the LDS list; everything else stays on topk.hip's kernel.
#include "common.hpp"

namespace wv {

constexpr int kWinBins = 32;                 // distance bins per window
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

template <int TPQ>
__host__ __device__ inline size_t rank2_lds_bytes_per_query(int k)
{
    size_t b = (size_t)kWinRows * (TPQ / 2) * 4;                 // table
    b += ((size_t)(k + TPQ) * 2 + 15) / 16 * 16;                 // stage
    b += (size_t)(kMaxBins2 + 1 + kWinBins + 4 + 3) / 4 * 4 * 4; // gbase, tot, misc
    return (b + 15) / 16 * 16;
}

// NC = distance-cache words (4 items each): items per thread C <= 4 * NC
template <int WORDS, int TPQ, int NC>
__device__ __forceinline__ void rank2_one_query(const uint64_t *__restrict__ dbT, const QCode<WORDS> &qc, int64_t N, int C,
                                                int nbins, int k, int64_t idx_offset, int32_t *__restrict__ idx_out,
                                                uint8_t *__restrict__ dist_out, uint32_t *__restrict__ cum_out,
                                                uint8_t *lds_raw, int t)
{
    constexpr int UNR = 8;
    constexpr int ROWB = TPQ * 2;                                // bytes per table row
    Rank2Lds L;
    L.table = reinterpret_cast<uint32_t *>(lds_raw);
    uint8_t *p = lds_raw + (size_t)kWinRows * ROWB;
    L.stage = reinterpret_cast<uint16_t *>(p);
    p += ((size_t)(k + TPQ) * 2 + 15) / 16 * 16;
    L.gbase = reinterpret_cast<uint32_t *>(p);
    L.tot = L.gbase + kMaxBins2 + 1;
    L.misc = L.tot + kWinBins;
    const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform
    constexpr int NW = TPQ / 64;                                 // waves per query

    // ---------------------------------------------------------------- phase 0: distances -> registers (bytes)
    const int first = t * C;
    const int nvalid = (int)min((int64_t)C, max((int64_t)0, N - (int64_t)first));
    uint32_t dc[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) dc[i] = 0xffffffffu;            // 255 = "no item": never inside a window
    uint32_t dmin = 255;
#pragma unroll
    for (int bi = 0; bi < NC * 4 / UNR; ++bi) {
        if (bi * UNR < C) {                                      // uniform
            QCode<WORDS> cur[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int r = min(bi * UNR + u, C - 1);
                const uint64_t *src = dbT + ((int64_t)r * TPQ + t) * WORDS;
                if constexpr (WORDS == 2) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(src);
                    cur[u].w[0] = (uint64_t)v.x | ((uint64_t)v.y << 32);
                    cur[u].w[1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
                } else {
                    cur[u].w[0] = src[0];
                }
            }
#pragma unroll
            for (int u4 = 0; u4 < UNR / 4; ++u4) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 4 * u4 + j;
                    uint32_t d = 0;
#pragma unroll
                    for (int w = 0; w < WORDS; ++w) d += (uint32_t)__popcll(cur[u].w[w] ^ qc.w[w]);
                    d = (bi * UNR + u < nvalid) ? d : 255u;      // select, not a branch
                    dmin = min(dmin, d);
                    word |= d << (8 * j);
                }
                dc[bi * (UNR / 4) + u4] = word;
            }
        }
    }
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
    const uint32_t cell_addr = (uint32_t)(t >> 1) * 4u;          // byte offset of this thread's dword inside a row
    const uint32_t cell_inc = 1u << (16 * (t & 1));
    const uint32_t cell_shift = 16u * (t & 1);
    char *tbl = reinterpret_cast<char *>(L.table);
    const uint32_t trash = (uint32_t)(k + t);

    for (;;) {
        const bool place = placed < (uint32_t)k;                 // uniform: false = count-only pass (cum requested)
        // ---- zero the table
        {
            uint4 *t4 = reinterpret_cast<uint4 *>(L.table);
            constexpr int n4 = kWinRows * ROWB / 16;
            for (int i = t; i < n4; i += TPQ) t4[i] = make_uint4(0, 0, 0, 0);
        }
        group_sync<TPQ>();
        // ---- count: one LDS add per item, into the row of its bin or into the dummy row.  Batches of 8 items (two
        // cache words): one uniform branch per batch, and the scheduler may not pile up more than a batch of LDS ops
#pragma unroll
        for (int bw = 0; bw < NC; bw += 2) {
            if (bw * 4 < C) {                                     // uniform; items >= C of the batch hold 255 -> dummy row
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (bw + (j >> 2) < NC) {
                        const uint32_t d = (dc[bw + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                        const uint32_t b = min(d - (uint32_t)lo, (uint32_t)kWinBins);
                        __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(tbl + b * ROWB + cell_addr), cell_inc,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        group_sync<TPQ>();
        // ---- per-bin totals (the waves of the group share the bins)
        for (int b = wv; b < kWinBins; b += NW) {
            uint32_t s;
            if constexpr (TPQ == 64) {
                s = (L.table[b * (TPQ / 2) + (lane >> 1)] >> cell_shift) & 0xffffu;
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(L.table + b * (TPQ / 2) + 2 * lane);
                s = (v.x & 0xffffu) + (v.x >> 16) + (v.y & 0xffffu) + (v.y >> 16);
            }
            s = wave_sum_u32(s);
            if (lane == 0) L.tot[b] = s;
        }
        group_sync<TPQ>();
        // ---- every wave: exclusive scan of the 32 totals (lane b holds bin b), then its own bins' thread scans
        uint32_t my_tot = lane < kWinBins ? L.tot[lane] : 0u;
        const uint32_t incl = wave_incl_scan_u32(my_tot);
        const uint32_t bin_base = placed + incl - my_tot;         // lane b: rows with distance < lo + b
        const uint32_t win_total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (wv == 0 && lane < kWinBins && lo + lane + 1 <= nbins) L.gbase[lo + lane + 1] = placed + incl;
        if (place) {
            for (int b = wv; b < kWinBins; b += NW) {
                const uint32_t base_b = (uint32_t)__builtin_amdgcn_readlane((int)bin_base, b);
                if constexpr (TPQ == 64) {
                    const uint32_t c = (L.table[b * (TPQ / 2) + (lane >> 1)] >> cell_shift) & 0xffffu;
                    const uint32_t excl = wave_incl_scan_u32(c) - c + base_b;
                    reinterpret_cast<uint16_t *>(L.table + b * (TPQ / 2))[lane] = (uint16_t)min(excl, 0xffffu);
                } else {
                    uint2 v = *reinterpret_cast<const uint2 *>(L.table + b * (TPQ / 2) + 2 * lane);
                    const uint32_t c0 = v.x & 0xffffu, c1 = v.x >> 16, c2 = v.y & 0xffffu, c3 = v.y >> 16;
                    const uint32_t s = c0 + c1 + c2 + c3;
                    const uint32_t e0 = wave_incl_scan_u32(s) - s + base_b;
                    const uint32_t e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2;
                    v.x = (e0 & 0xffffu) | (e1 << 16);
                    v.y = (e2 & 0xffffu) | (e3 << 16);
                    *reinterpret_cast<uint2 *>(L.table + b * (TPQ / 2) + 2 * lane) = v;
                }
            }
            // dummy row: every cell starts at k, so whatever it returns is >= k (trash); k + 128 < 65536 (host check)
            if (wv == NW - 1) {
                const uint32_t kk = (uint32_t)k | ((uint32_t)k << 16);
                for (int i = lane; i < TPQ / 2; i += 64) L.table[kWinBins * (TPQ / 2) + i] = kk;
            }
            group_sync<TPQ>();
            // ---- placement: returning LDS add = this item's rank, item number into the LDS list (or the trash slot)
#pragma unroll
            for (int bw = 0; bw < NC; bw += 2) {
                if (bw * 4 < C) {                                 // uniform
                    uint32_t old[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        old[j] = 0;
                        if (bw + (j >> 2) < NC) {
                            const uint32_t d = (dc[bw + (j >> 2)] >> (8 * (j & 3))) & 0xffu;
                            const uint32_t b = min(d - (uint32_t)lo, (uint32_t)kWinBins);
                            old[j] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(tbl + b * ROWB + cell_addr), cell_inc,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (bw + (j >> 2) < NC) {
                            const uint32_t pos = (old[j] >> cell_shift) & 0xffffu;
                            L.stage[min(pos, trash)] = (uint16_t)(first + bw * 4 + j);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        placed += win_total;
        lo += kWinBins;
        // uniform exit: the list is complete and nobody asked for the full histogram, or no bins are left
        if (lo >= nbins || (placed >= (uint32_t)k && !cum_out)) break;
        group_sync<TPQ>();                                        // table is re-zeroed next
    }
    group_sync<TPQ>();
    // bins beyond the last window processed hold every row (the loop only stops early once placed >= k)
    // ---- cumulative histogram of all rows (cum[b] = rows with distance < b), for the sharded search
    if (cum_out)
        for (int b = t; b <= nbins; b += TPQ) cum_out[b] = min(L.gbase[b], (uint32_t)N);
    // ---- the ranked list leaves with 16-byte stores
    {
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
    // ---- distance row from the bin boundaries: dist[p] = b with gbase[b] <= p < gbase[b+1]; 16 positions per thread
    if (dist_out) {
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

// dbT image for TPQ threads per query: dbT[r][t] = code[t*C + r], C = ceil(N / TPQ); rows beyond N are zero
template <int WORDS>
__global__ __launch_bounds__(256) void k_transpose_db2(const uint64_t *__restrict__ db, uint64_t *__restrict__ dbT,
                                                       int64_t N, int C, int tpq)
{
    const int64_t total = (int64_t)C * tpq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / tpq;
        const int t = (int)(i - r * tpq);
        const int64_t item = (int64_t)t * C + r;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) dbT[i * WORDS + w] = item < N ? db[item * WORDS + w] : 0ull;
    }
}

// minimum waves per SIMD the register allocation has to leave room for (the LDS footprint admits at least as many)
constexpr int rank2_min_waves(int nc, int tpq) { return nc <= 8 ? (tpq == 64 ? 8 : 7) : (nc <= 16 ? 6 : (nc <= 25 ? 5 : 4)); }

template <int WORDS, int TPQ, int NC>
__global__ __launch_bounds__(256, rank2_min_waves(NC, TPQ)) void k_rank_window(const uint64_t *__restrict__ q, const uint64_t *__restrict__ dbT,
                                                     int32_t *__restrict__ idx, uint8_t *__restrict__ dist, int Q,
                                                     int64_t N, int C, int nbins, int k, int64_t idx_offset,
                                                     uint32_t *__restrict__ cum, int lds_per_query)
{
    extern __shared__ uint4 lds4[];
    constexpr int QPW = 256 / TPQ;                              // queries per workgroup
    const int g = threadIdx.x / TPQ, t = threadIdx.x % TPQ;
    const int qi = blockIdx.x * QPW + g;
    if (QPW > 1 && qi >= Q) return;                              // whole waves only (TPQ == 64): no barrier is skipped
    QCode<WORDS> qc;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) {
        const uint64_t v = q[(int64_t)qi * WORDS + w];
        // the query is uniform over its group: keep it in SGPRs
        qc.w[w] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    }
    rank2_one_query<WORDS, TPQ, NC>(dbT, qc, N, C, nbins, k, idx_offset, idx + (int64_t)qi * k,
                                    dist ? dist + (int64_t)qi * k : nullptr, cum ? cum + (int64_t)qi * (nbins + 1) : nullptr,
                                    reinterpret_cast<uint8_t *>(lds4) + (size_t)g * lds_per_query, t);
}

// ------------------------------------------------------------------------------------------ host side
// which thread count per query a shape takes: 0 = not covered by this kernel
int rank2_tpq(int Q, int64_t N, int k)
{
    const char *force = getenv("WV_TOPK_V2");                   // "0": off, "64" / "256": pin the variant (tests, tuning)
    if (force && force[0] == '0') return 0;
    if (N >= 65536 || k + 128 >= 65536) return 0;
    const bool fits256 = ceil_div(N, 256) <= 128 && rank2_lds_bytes_per_query<256>(k) <= 100 * 1024;
    const bool fits64 = ceil_div(N, 64) <= 128 && 4 * rank2_lds_bytes_per_query<64>(k) <= 100 * 1024;
    if (force && atoi(force) == 64) return fits64 ? 64 : 0;
    if (force && atoi(force) == 256) return fits256 ? 256 : 0;
    // one wave per query pays when each query has little work and there are enough queries to fill the chip
    if (fits64 && N <= 4096 && Q >= 4096) return 64;
    return fits256 ? 256 : (fits64 ? 64 : 0);
}

size_t rank2_image_bytes(int64_t N, int words, int tpq)
{
    return (size_t)ceil_div(N, tpq) * tpq * words * sizeof(uint64_t);
}

int rank2_prepare(const uint64_t *db, void *img, int64_t N, int words, int tpq, hipStream_t st)
{
    const int C = (int)ceil_div(N, tpq);
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div((int64_t)C * tpq, 256), 4096);
    if (words == 1)
        hipLaunchKernelGGL((k_transpose_db2<1>), dim3(grid), dim3(256), 0, st, db, (uint64_t *)img, N, C, tpq);
    else
        hipLaunchKernelGGL((k_transpose_db2<2>), dim3(grid), dim3(256), 0, st, db, (uint64_t *)img, N, C, tpq);
    WV_CHECK_LAUNCH("k_transpose_db2");
    return WV_OK;
}

template <int WORDS, int TPQ, int NC>
static int launch_rank2_nc(const uint64_t *q, const uint64_t *dbT, int32_t *idx, uint8_t *dist, int Q, int64_t N, int C,
                           int nbins, int k, int64_t idx_offset, uint32_t *cum, hipStream_t st)
{
    constexpr int QPW = 256 / TPQ;
    const size_t per_q = rank2_lds_bytes_per_query<TPQ>(k), lds = per_q * QPW;
    auto kern = k_rank_window<WORDS, TPQ, NC>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "rank_window: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(Q, QPW)), dim3(256), lds, st, q, dbT, idx, dist, Q, N, C, nbins, k,
                       idx_offset, cum, (int)per_q);
    WV_CHECK_LAUNCH("k_rank_window");
    return WV_OK;
}

template <int WORDS, int TPQ>
static int launch_rank2_t(const uint64_t *q, const uint64_t *dbT, int32_t *idx, uint8_t *dist, int Q, int64_t N, int nbins,
                          int k, int64_t idx_offset, uint32_t *cum, hipStream_t st)
{
    const int C = (int)ceil_div(N, TPQ);
#define WV_R2(NCW) return launch_rank2_nc<WORDS, TPQ, NCW>(q, dbT, idx, dist, Q, N, C, nbins, k, idx_offset, cum, st)
    if (C <= 16) WV_R2(4);
    if (C <= 32) WV_R2(8);
    if (C <= 64) WV_R2(16);
    if (C <= 100) WV_R2(25);
    WV_R2(32);
#undef WV_R2
}

// dbT must be the image for `tpq` threads per query (rank2_prepare)
int rank2_launch(const uint64_t *q, const uint64_t *dbT, int32_t *idx, uint8_t *dist, int Q, int64_t N, int nbits, int k,
                 int64_t idx_offset, uint32_t *cum, int tpq, hipStream_t st)
{
    const int nbins = nbits + 1, words = (nbits + 63) / 64;
    if (words == 1) {
        if (tpq == 64) return launch_rank2_t<1, 64>(q, dbT, idx, dist, Q, N, nbins, k, idx_offset, cum, st);
        return launch_rank2_t<1, 256>(q, dbT, idx, dist, Q, N, nbins, k, idx_offset, cum, st);
    }
    if (tpq == 64) return launch_rank2_t<2, 64>(q, dbT, idx, dist, Q, N, nbins, k, idx_offset, cum, st);
    return launch_rank2_t<2, 256>(q, dbT, idx, dist, Q, N, nbins, k, idx_offset, cum, st);
}

}  // namespace wv

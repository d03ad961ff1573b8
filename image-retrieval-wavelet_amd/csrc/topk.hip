// Fused Hamming distance + stable top-k ranking, list merge and average precision for gfx950.
//
// Reference: per query  hamm = 0.5*(B - q @ r.T); indices = torch.argsort(hamm)[:topk]
// (accuracy_calculator.py:219-223) -- an O(N log N) comparison sort of a key with only B+1
// distinct values.  Here: an exact counting sort in O(N), ties broken by ascending database
// index (= torch.argsort(stable=True)).
//
// One workgroup (256 threads) ranks one query.  Thread t owns the CONTIGUOUS item range
// [t*C, (t+1)*C), C = ceil(N/256), and a private column hist[bin][t] of an LDS table, so that
//   phase 1  hist[d][t] += 1                     (conflict-free: bank = t mod 32 for every bin)
//   scan     offs[d][t] = sum_{b<d} tot[b] + sum_{t'<t} hist[d][t']   (exclusive, bin-major)
//   phase 2  pos = offs[d][t]++  -> idx[pos] = item      (only for d <= threshold bin T)
// reproduces the stable order with no atomics, no sort and no inter-lane matching.  The
// database is pre-transposed once per call into dbT[r][t] = code[t*C + r] so that the
// per-thread contiguous ranges are read with fully coalesced loads.  The distance row is not
// scattered at all: it is regenerated from the bin boundaries (sorted => run-length).
#include "common.hpp"
#include "ap_walk.hpp"
#include <type_traits>

namespace wv {

constexpr int kTopkThreads = 256;

// second-generation kernel (rank2.hip): windowed count table, branch-free, LDS-assembled list
int rank2_tpq(int Q, int64_t N, int k);
size_t rank2_image_bytes(int64_t N, int words, int tpq);
int rank2_prepare(const uint64_t *db, void *img, int64_t N, int words, int tpq, hipStream_t st);
size_t rank2_labels_bytes(int64_t N, int lwords);
int rank2_labels_prepare(const uint64_t *dblab, void *cls, int64_t N, int lwords, hipStream_t st);
// lab_img (the class-major label bit matrix) / qlab / ap / nrel: average precision of the list (wv_hamming_map_at_k); all NULL otherwise
int rank2_launch(const uint64_t *q, const void *img, int32_t *idx, uint16_t *rows16, uint8_t *dist, int Q, int64_t N, int nbits,
                 int k, int64_t idx_offset, uint32_t *cum, int tpq, hipStream_t st, const void *lab_img = nullptr,
                 const uint64_t *qlab = nullptr, float *ap = nullptr, int32_t *nrel = nullptr, uint64_t *relbits = nullptr,
                 int64_t relbits_ld = 0, int64_t cum_ld = 0, int lwords = 1);
// the one-wave-per-query image exists for databases (shards) of at most this many rows
constexpr int64_t kImg64MaxRows = 64 * 64;
// images of the windowed kernel exist for databases it can take at all (16-bit item numbers, <= 128 items per thread)
constexpr int64_t kImg256MaxRows = 256 * 128;
// prepared ranking blob: [first-generation column image][windowed kernel, 256 threads/query][.., 64 threads/query]
static inline size_t old_image_bytes(int64_t N, int words) { return (size_t)ceil_div(N, kTopkThreads) * kTopkThreads * words * sizeof(uint64_t); }
static inline size_t r2_off256(int64_t N, int words) { return (size_t)align_up((int64_t)old_image_bytes(N, words), 256); }
static inline size_t r2_off64(int64_t N, int words)
{
    return r2_off256(N, words) + (N <= kImg256MaxRows ? (size_t)align_up((int64_t)rank2_image_bytes(N, words, 256), 256) : 0);
}
__device__ int g_topk_dbg = 0;   // WV_TOPK_DBG (timing experiments only): 1 = no list stores, 2 = skip phase 2, 4 = skip phase 1
constexpr int kMaxBins = 130;  // nbits <= 128 (+1 bin for the padding value of ragged shard lists)

// ------------------------------------------------------------------------ item sources
// CodeSource: items are database codes, distance = popcount(q ^ code), id = row + offset
template <int WORDS>
struct CodeSource {
    const uint64_t *dbT;  // [C][256][WORDS]
    uint64_t qw[WORDS];
    int64_t idx_offset;
    struct Raw {
        uint64_t w[WORDS];
    };
    __device__ __forceinline__ Raw fetch(int r, int t, int64_t) const
    {
        const uint64_t *p = dbT + ((int64_t)r * kTopkThreads + t) * WORDS;
        Raw c;
        if constexpr (WORDS == 2) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p);
            c.w[0] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            c.w[1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        } else {
#pragma unroll
            for (int w = 0; w < WORDS; ++w) c.w[w] = p[w];
        }
        return c;
    }
    __device__ __forceinline__ int dist(const Raw &c) const
    {
        int d = 0;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) d += __popcll(c.w[w] ^ qw[w]);
        return d;
    }
    __device__ __forceinline__ int32_t id(int64_t item) const { return (int32_t)(item + idx_offset); }
};

// RowSource: items are the entries of one stored distance-matrix row
struct RowSource {
    const uint8_t *row;
    int64_t n;
    using Raw = int;
    __device__ __forceinline__ Raw fetch(int, int, int64_t item) const { return item < n ? row[item] : 0; }
    __device__ __forceinline__ int dist(const Raw &c) const { return c; }
    __device__ __forceinline__ int32_t id(int64_t item) const { return (int32_t)item; }
};

// ------------------------------------------------------------------------ the ranking core
// LDS: hist[nbins][256] (u32, or u16 pairs when n_items < 65536), tot[kMaxBins], base[kMaxBins + 1], misc
//
// U16 = true packs two threads' counters into one dword (thread t uses half t & 1 of dword t >> 1): half
// the LDS, twice the resident workgroups.  Counters and offsets stay below 65536 because n_items does.
template <bool U16>
struct Hist {
    uint32_t *h;
    __device__ __forceinline__ uint32_t *slot(int bin, int t) const
    {
        return U16 ? h + bin * (kTopkThreads / 2) + (t >> 1) : h + bin * kTopkThreads + t;
    }
    __device__ __forceinline__ void add(int bin, int t) const
    {
        __hip_atomic_fetch_add(slot(bin, t), U16 ? (1u << (16 * (t & 1))) : 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ uint32_t fetch_inc(int bin, int t) const
    {
        const uint32_t old = __hip_atomic_fetch_add(slot(bin, t), U16 ? (1u << (16 * (t & 1))) : 1u,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return U16 ? (old >> (16 * (t & 1))) & 0xffffu : old;
    }
    // the four counters of threads 4*lane .. 4*lane+3 of one bin
    __device__ __forceinline__ uint4 load4(int bin, int lane) const
    {
        if (U16) {
            const uint2 v = *reinterpret_cast<const uint2 *>(h + bin * (kTopkThreads / 2) + 2 * lane);
            return make_uint4(v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16);
        }
        return *reinterpret_cast<const uint4 *>(h + bin * kTopkThreads + 4 * lane);
    }
    __device__ __forceinline__ void store4(int bin, int lane, uint4 o) const
    {
        if (U16)
            *reinterpret_cast<uint2 *>(h + bin * (kTopkThreads / 2) + 2 * lane) = make_uint2(o.x | (o.y << 16), o.z | (o.w << 16));
        else
            *reinterpret_cast<uint4 *>(h + bin * kTopkThreads + 4 * lane) = o;
    }
    static __host__ __device__ size_t words(int nbins) { return (size_t)nbins * (U16 ? kTopkThreads / 2 : kTopkThreads); }
};

// STAGED: the ranked indices are first placed in an LDS row (scattered 4-byte LDS writes), then copied out
// with 16-byte coalesced stores -- instead of k scattered 4-byte global stores per query.
template <typename Source, bool U16, bool STAGED>
__device__ __forceinline__ void rank_one_query(const Source &src, int64_t n_items, int C, int nbins,
                                               int k, int32_t *__restrict__ idx_out,
                                               uint8_t *__restrict__ dist_out, uint32_t *lds,
                                               uint32_t *__restrict__ cum_out = nullptr)
{
    Hist<U16> hist{lds};
    uint32_t *tot = lds + Hist<U16>::words(nbins);     // kMaxBins
    uint32_t *base = tot + kMaxBins;                   // kMaxBins + 1
    uint32_t *misc = base + kMaxBins + 1;              // [0] = threshold bin T
    // staging row, 16-byte aligned: starts at the rounded-down word count used by rank_lds_bytes()
    int32_t *stage = reinterpret_cast<int32_t *>(lds + (Hist<U16>::words(nbins) + kMaxBins + kMaxBins + 1 + 4 + 3) / 4 * 4);
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();

    {   // zero the table with 16-byte stores
        uint4 *h4 = reinterpret_cast<uint4 *>(lds);
        const int n4 = (int)(Hist<U16>::words(nbins) / 4);
        for (int i = tid; i < n4; i += kTopkThreads) h4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // ---- phase 1: private-column histogram.  A column belongs to one thread, so the LDS adds never
    // contend; nothing waits on them.  Distances of the next UNR items are computed while they drain.
    const int64_t first = (int64_t)tid * C;
    // items of this thread that exist (only the last threads of the last column are short)
    const int nvalid = (int)min((int64_t)C, max((int64_t)0, n_items - first));
    constexpr int UNR = 16;
    using Raw = typename Source::Raw;
    const int nfull = C / UNR;
    const int dbg = g_topk_dbg;
    // Columns of at most kCacheItems items keep their distances (one byte each) in registers between the
    // two phases, so phase 2 neither reloads the codes nor recomputes the popcounts.
    constexpr int kCacheItems = 128, kCacheBatches = kCacheItems / UNR;
    const bool cached = C <= kCacheItems;            // uniform
    uint32_t dcache[kCacheItems / 4];
    if (cached) {
        if (!(dbg & 4)) {
#pragma unroll
            for (int bi = 0; bi < kCacheBatches; ++bi) {
                if (bi * UNR < C) {                   // uniform
                    Raw cur[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) cur[u] = src.fetch(min(bi * UNR + u, C - 1), tid, first + bi * UNR + u);
#pragma unroll
                    for (int u4 = 0; u4 < UNR / 4; ++u4) {
                        uint32_t word = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int u = 4 * u4 + j;
                            const int d = src.dist(cur[u]);
                            word |= (uint32_t)d << (8 * j);
                            if (bi * UNR + u < nvalid) hist.add(d, tid);
                        }
                        dcache[bi * (UNR / 4) + u4] = word;
                    }
                }
            }
        }
    } else if (!(dbg & 4)) {
        Raw cur[UNR], nxt[UNR];
        if (nfull > 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) cur[u] = src.fetch(u, tid, first + u);
        }
        for (int bi = 0; bi < nfull; ++bi) {
            const int r = bi * UNR;
            if (bi + 1 < nfull) {   // the next batch's loads are in flight while this one is counted
#pragma unroll
                for (int u = 0; u < UNR; ++u) nxt[u] = src.fetch(r + UNR + u, tid, first + r + UNR + u);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (r + u < nvalid) hist.add(src.dist(cur[u]), tid);
#pragma unroll
            for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
        }
        for (int r = nfull * UNR; r < nvalid; ++r) hist.add(src.dist(src.fetch(r, tid, first + r)), tid);
    }
    __syncthreads();

    // ---- per-bin totals (wave w takes bins w, w+4, ...)
    for (int b = wv; b < nbins; b += kTopkThreads / 64) {
        const uint4 v = hist.load4(b, lane);
        const uint32_t s = wave_sum_u32(v.x + v.y + v.z + v.w);
        if (lane == 0) tot[b] = s;
    }
    __syncthreads();

    // ---- exclusive scan over bins (wave 0; up to 3 bins per lane covers 192 >= 130 bins)
    if (wv == 0) {
        uint32_t t0 = 0, t1 = 0, t2 = 0;
        const int b0 = 3 * lane;
        if (b0 < nbins) t0 = tot[b0];
        if (b0 + 1 < nbins) t1 = tot[b0 + 1];
        if (b0 + 2 < nbins) t2 = tot[b0 + 2];
        const uint32_t incl = wave_incl_scan_u32(t0 + t1 + t2);
        const uint32_t excl = incl - (t0 + t1 + t2);
        if (b0 < nbins) base[b0] = excl;
        if (b0 + 1 < nbins) base[b0 + 1] = excl + t0;
        if (b0 + 2 < nbins) base[b0 + 2] = excl + t0 + t1;
        if (lane == 63) base[nbins] = incl;  // = n_items
        // threshold bin: first bin whose inclusive count reaches k
        int cand = nbins;  // sentinel
        if (b0 < nbins && excl + t0 >= (uint32_t)k) cand = b0;
        else if (b0 + 1 < nbins && excl + t0 + t1 >= (uint32_t)k) cand = b0 + 1;
        else if (b0 + 2 < nbins && excl + t0 + t1 + t2 >= (uint32_t)k) cand = b0 + 2;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) cand = min(cand, __shfl_xor(cand, d, 64));
        if (lane == 0) misc[0] = (uint32_t)min(cand, nbins - 1);
    }
    __syncthreads();
    const int T = (int)misc[0];
    // cumulative histogram of ALL items (cum[b] = #items with distance < b): lets a sharded search derive the
    // global threshold from one small all-reduce instead of exchanging full-length lists
    if (cum_out)
        for (int b = tid; b <= nbins; b += kTopkThreads) cum_out[b] = base[b];

    // ---- per-bin exclusive scan over threads, plus the bin base -> starting output offsets
    for (int b = wv; b <= T; b += kTopkThreads / 64) {
        const uint4 v = hist.load4(b, lane);
        const uint32_t s = v.x + v.y + v.z + v.w;
        const uint32_t excl = wave_incl_scan_u32(s) - s + base[b];
        uint4 o;
        o.x = excl;
        o.y = excl + v.x;
        o.z = o.y + v.y;
        o.w = o.z + v.z;
        hist.store4(b, lane, o);
    }
    __syncthreads();

    // ---- phase 2: stable placement of the items in bins <= T (returning LDS adds, UNR in flight)
    if (cached) {
        if (!(dbg & 2)) {
#pragma unroll
            for (int bi = 0; bi < kCacheBatches; ++bi) {
                if (bi * UNR < C) {
                    uint32_t pos[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int d = (dcache[bi * (UNR / 4) + u / 4] >> (8 * (u & 3))) & 0xff;
                        pos[u] = 0xffffffffu;
                        if (bi * UNR + u < nvalid && d <= T) pos[u] = hist.fetch_inc(d, tid);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        if (pos[u] < (uint32_t)k && !(dbg & 1)) {
                            if (STAGED) stage[pos[u]] = src.id(first + bi * UNR + u);
                            else idx_out[pos[u]] = src.id(first + bi * UNR + u);
                        }
                }
            }
        }
    } else if (!(dbg & 2)) {
        Raw cur[UNR], nxt[UNR];
        if (nfull > 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) cur[u] = src.fetch(u, tid, first + u);
        }
        for (int bi = 0; bi < nfull; ++bi) {
            const int r = bi * UNR;
            if (bi + 1 < nfull) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) nxt[u] = src.fetch(r + UNR + u, tid, first + r + UNR + u);
            }
            uint32_t pos[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int d = src.dist(cur[u]);
                pos[u] = 0xffffffffu;
                if (r + u < nvalid && d <= T) pos[u] = hist.fetch_inc(d, tid);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (pos[u] < (uint32_t)k && !(dbg & 1)) {
                    if (STAGED) stage[pos[u]] = src.id(first + r + u);
                    else idx_out[pos[u]] = src.id(first + r + u);
                }
#pragma unroll
            for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
        }
        for (int r = nfull * UNR; r < nvalid; ++r) {
            const int64_t item = first + r;
            const int d = src.dist(src.fetch(r, tid, item));
            if (d <= T) {
                const uint32_t pos = hist.fetch_inc(d, tid);
                if (pos < (uint32_t)k && !(dbg & 1)) {
                    if (STAGED) stage[pos] = src.id(item);
                    else idx_out[pos] = src.id(item);
                }
            }
        }
    }
    if (STAGED) {   // coalesced copy-out of the ranked row
        __syncthreads();
        if ((k & 3) == 0 && (reinterpret_cast<uintptr_t>(idx_out) & 15) == 0) {
            const int4 *s4 = reinterpret_cast<const int4 *>(stage);
            int4 *o4 = reinterpret_cast<int4 *>(idx_out);
            for (int i = tid; i < k / 4; i += kTopkThreads) o4[i] = s4[i];
        } else {
            for (int i = tid; i < k; i += kTopkThreads) idx_out[i] = stage[i];
        }
    }

    // ---- distance row from the bin boundaries: dist[p] = b  with  base[b] <= p < base[b+1].
    // Each thread owns a run of positions: one binary search, then it walks the boundaries; 4 bytes per store.
    if (dist_out) {
        const int chunk = ((k + kTopkThreads - 1) / kTopkThreads + 3) & ~3;
        const int p0 = tid * chunk;
        if (p0 < k) {
            int lo = 0, hi = nbins;  // invariant: base[lo] <= p0 < base[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (base[mid] <= (uint32_t)p0) lo = mid;
                else hi = mid;
            }
            int bin = lo;
            uint32_t next = base[bin + 1];
            const bool aligned = (reinterpret_cast<uintptr_t>(dist_out) & 3) == 0;
            for (int i = 0; i < chunk; i += 4) {
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t p = (uint32_t)(p0 + i + j);
                    while (p >= next && bin + 1 < nbins) { ++bin; next = base[bin + 1]; }
                    word |= (uint32_t)bin << (8 * j);
                }
                if (aligned && p0 + i + 4 <= k) {
                    *reinterpret_cast<uint32_t *>(dist_out + p0 + i) = word;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (p0 + i + j < k) dist_out[p0 + i + j] = (uint8_t)(word >> (8 * j));
                }
            }
        }
    }
}

constexpr int kStageMaxK = 8192;   // ranked rows up to this length are staged in LDS (32 KB)
static inline size_t rank_fixed_words(int nbins, bool u16)
{
    const size_t h = (size_t)nbins * (u16 ? kTopkThreads / 2 : kTopkThreads);
    return (h + kMaxBins + kMaxBins + 1 + 4 + 3) / 4 * 4;   // 16-byte aligned start of the staging row
}
// Measured on MI355X (c1 shape): staging costs a resident workgroup per CU (53 KB vs 34 KB of LDS) and
// runs 163 us vs 117 us with direct scattered stores -- the kernel is latency-bound, so occupancy wins.
// Kept selectable (WV_TOPK_STAGE=1) for shapes where the table is small.
static inline bool rank_staged(int k, int nbins, bool u16)
{
    static const bool enabled = ::wv::tune("WV_TOPK_STAGE") && ::wv::tune("WV_TOPK_STAGE")[0] == '1';
    return enabled && k <= kStageMaxK && (rank_fixed_words(nbins, u16) + (size_t)k + 4) * 4 <= (size_t)kMaxLdsBytes - 4096;
}
static inline size_t rank_lds_bytes(int nbins, bool u16, int k)
{
    return (rank_fixed_words(nbins, u16) + (rank_staged(k, nbins, u16) ? (size_t)(k + 3) / 4 * 4 : 0)) * sizeof(uint32_t);
}
static inline bool rank_u16(int64_t n_items) { return n_items < 65536; }

// ------------------------------------------------------------------------ kernels
template <int WORDS>
__global__ __launch_bounds__(kTopkThreads) void k_transpose_db(const uint64_t *__restrict__ db,
                                                               uint64_t *__restrict__ dbT,
                                                               int64_t N, int C)
{
    const int64_t total = (int64_t)C * kTopkThreads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / kTopkThreads;
        const int t = (int)(i - r * kTopkThreads);
        const int64_t item = (int64_t)t * C + r;
#pragma unroll
        for (int w = 0; w < WORDS; ++w) dbT[i * WORDS + w] = item < N ? db[item * WORDS + w] : 0ull;
    }
}

template <int WORDS, bool U16, bool STAGED>
__global__ __launch_bounds__(kTopkThreads) void k_hamming_topk(const uint64_t *__restrict__ q,
                                                               const uint64_t *__restrict__ dbT,
                                                               int32_t *__restrict__ idx,
                                                               uint8_t *__restrict__ dist, int64_t N,
                                                               int C, int nbins, int k,
                                                               int64_t idx_offset, uint32_t *__restrict__ cum)
{
    extern __shared__ uint4 lds4[];
    const int qi = blockIdx.x;
    CodeSource<WORDS> src;
    src.dbT = dbT;
    src.idx_offset = idx_offset;
#pragma unroll
    for (int w = 0; w < WORDS; ++w) src.qw[w] = q[(int64_t)qi * WORDS + w];
    rank_one_query<CodeSource<WORDS>, U16, STAGED>(src, N, C, nbins, k, idx + (int64_t)qi * k,
                                           dist ? dist + (int64_t)qi * k : nullptr, reinterpret_cast<uint32_t *>(lds4),
                                           cum ? cum + (int64_t)qi * (nbins + 1) : nullptr);
}

template <bool U16, bool STAGED>
__global__ __launch_bounds__(kTopkThreads) void k_rank_from_dist(const uint8_t *__restrict__ dmat,
                                                                 int64_t ld, int64_t N,
                                                                 int32_t *__restrict__ idx,
                                                                 uint8_t *__restrict__ dist, int k,
                                                                 int C, int nbins)
{
    extern __shared__ uint4 lds4[];
    const int qi = blockIdx.x;
    RowSource src;
    src.row = dmat + (int64_t)qi * ld;
    src.n = N;
    rank_one_query<RowSource, U16, STAGED>(src, N, C, nbins, k, idx + (int64_t)qi * k, dist ? dist + (int64_t)qi * k : nullptr,
                                   reinterpret_cast<uint32_t *>(lds4));
}

// ------------------------------------------------------------------------ merge of sorted shard lists
// The G input lists of a query are each sorted by (distance, index) and come from contiguous row shards in
// rank order, so the merged order is: by distance bin, then by shard, then by position inside the shard's
// run of that distance.  Run boundaries come from binary searches on the sorted distance rows (no per-item
// histogram, no atomics); every entry's output position is  p + delta[g][d]  with one LDS lookup, entries are
// read in coalesced order and land in contiguous runs.
__global__ __launch_bounds__(256) void k_merge_sorted(const int32_t *__restrict__ idx_in,
                                                      const uint8_t *__restrict__ dist_in, int G, int Q, int kin,
                                                      int32_t *__restrict__ idx_out,
                                                      uint8_t *__restrict__ dist_out, int k, int nbins)
{
    extern __shared__ uint4 lds4[];
    int32_t *start = reinterpret_cast<int32_t *>(lds4);           // [G][nbins + 1]: first position with dist >= b
    int32_t *delta = start + G * (nbins + 1);                     // [G][nbins]
    uint32_t *base = reinterpret_cast<uint32_t *>(delta + G * nbins);   // [nbins + 1]
    const int qi = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t shard_stride = (int64_t)Q * kin;
    const uint8_t *dq = dist_in + (int64_t)qi * kin;
    const int32_t *iq = idx_in + (int64_t)qi * kin;

    for (int u = tid; u < G * (nbins + 1); u += 256) {
        const int g = u / (nbins + 1), b = u - g * (nbins + 1);
        const uint8_t *row = dq + g * shard_stride;
        int lo = 0, hi = kin;                                     // first p with row[p] >= b
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)row[mid] < b) lo = mid + 1;
            else hi = mid;
        }
        start[u] = lo;
    }
    __syncthreads();
    if (wv == 0) {   // totals per bin and their exclusive scan (up to 3 bins per lane)
        uint32_t t[3] = {0, 0, 0};
        const int b0 = 3 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (b0 + j < nbins)
                for (int g = 0; g < G; ++g) t[j] += (uint32_t)(start[g * (nbins + 1) + b0 + j + 1] - start[g * (nbins + 1) + b0 + j]);
        const uint32_t incl = wave_incl_scan_u32(t[0] + t[1] + t[2]);
        const uint32_t excl = incl - (t[0] + t[1] + t[2]);
        if (b0 < nbins) base[b0] = excl;
        if (b0 + 1 < nbins) base[b0 + 1] = excl + t[0];
        if (b0 + 2 < nbins) base[b0 + 2] = excl + t[0] + t[1];
    }
    __syncthreads();
    for (int b = tid; b < nbins; b += 256) {
        uint32_t run = base[b];
        for (int g = 0; g < G; ++g) {
            const int s0 = start[g * (nbins + 1) + b], s1 = start[g * (nbins + 1) + b + 1];
            delta[g * nbins + b] = (int32_t)run - s0;
            run += (uint32_t)(s1 - s0);
        }
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const uint8_t *row = dq + g * shard_stride;
        const int32_t *irow = iq + g * shard_stride;
        for (int p = tid; p < kin; p += 256) {
            const int d = row[p];
            if (d >= nbins) continue;                             // defensive: not a bin this call was sized for
            const int pos = p + delta[g * nbins + d];
            if (pos < k) {
                idx_out[(int64_t)qi * k + pos] = irow[p];
                if (dist_out) dist_out[(int64_t)qi * k + pos] = (uint8_t)d;
            }
        }
    }
}

// ------------------------------------------------------------------------ average precision
// AP over a ranked list (accuracy_calculator.py:222-229): hits at 1-based ranks r_1 < r_2 < ...
// give AP = mean_j (j / r_j); relevance = labels share a bit (label_comparison_fn :31-37).
template <int LW>
__global__ __launch_bounds__(256) void k_map_at_k(const int32_t *__restrict__ idx, int64_t ld, int k,
                                                  const uint64_t *__restrict__ qlab,
                                                  const uint64_t *__restrict__ dblab, int lwords,
                                                  float *__restrict__ ap, int32_t *__restrict__ nrel)
{
    __shared__ uint32_t wave_cnt[4];
    __shared__ double wave_sum[4];
    const int qi = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int32_t *list = idx + (int64_t)qi * ld;
    const int lw = LW > 0 ? LW : lwords;
    uint64_t ql[LW > 0 ? LW : 1];
    if constexpr (LW > 0) {
#pragma unroll
        for (int w = 0; w < LW; ++w) ql[w] = qlab[(int64_t)qi * LW + w];
    }
    uint32_t running = 0;
    double acc = 0.0;
    for (int p0 = 0; p0 < k; p0 += 256) {
        const int p = p0 + tid;
        bool rel = false;
        if (p < k) {
            const int32_t id = list[p];
            if (id >= 0) {
                const uint64_t *dl = dblab + (int64_t)id * lw;
                if constexpr (LW > 0) {
                    uint64_t any = 0;
#pragma unroll
                    for (int w = 0; w < LW; ++w) any |= dl[w] & ql[w];
                    rel = any != 0;
                } else {
                    uint64_t any = 0;
                    for (int w = 0; w < lw; ++w) any |= dl[w] & qlab[(int64_t)qi * lw + w];
                    rel = any != 0;
                }
            }
        }
        const uint64_t mask = __ballot(rel);
        if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = running;
        for (int w2 = 0; w2 < wv; ++w2) before += wave_cnt[w2];
        const uint32_t block_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (rel) {
            const uint32_t j = before + (uint32_t)mbcnt(mask) + 1;  // this is the j-th hit
            acc += (double)((float)j / (float)(p + 1));             // fp32 quotient like the reference
        }
        running += block_total;
        __syncthreads();
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) wave_sum[wv] = acc;
    __syncthreads();
    if (tid == 0) {
        const double s = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        ap[qi] = running ? (float)(s / (double)running) : 0.0f;
        if (nrel) nrel[qi] = (int32_t)running;
    }
}

// Running hit count along a ranked list: hits[q][p] = number of relevant entries among list[0..p]
// (relevance as in k_map_at_k).  Precision@p, recall@p, R-precision and the precision/recall curves of
// accuracy_calculator.py:131-181,235-273 are ratios of these counts.
__global__ __launch_bounds__(256) void k_hit_prefix(const int32_t *__restrict__ idx, int k,
                                                    const uint64_t *__restrict__ qlab,
                                                    const uint64_t *__restrict__ dblab, int lwords,
                                                    uint32_t *__restrict__ hits)
{
    __shared__ uint32_t wave_cnt[4];
    const int qi = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int32_t *list = idx + (int64_t)qi * k;
    uint32_t running = 0;
    for (int p0 = 0; p0 < k; p0 += 256) {
        const int p = p0 + tid;
        bool rel = false;
        if (p < k) {
            const int32_t id = list[p];
            if (id >= 0) {
                uint64_t any = 0;
                for (int w = 0; w < lwords; ++w) any |= dblab[(int64_t)id * lwords + w] & qlab[(int64_t)qi * lwords + w];
                rel = any != 0;
            }
        }
        const uint64_t mask = __ballot(rel);
        if (lane == 0) wave_cnt[wv] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t before = running;
        for (int w2 = 0; w2 < wv; ++w2) before += wave_cnt[w2];
        if (p < k) hits[(int64_t)qi * k + p] = before + (uint32_t)mbcnt(mask) + (rel ? 1u : 0u);
        running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// One wave: was every shard's prefix long enough for query qi?  T = the query's global k-th distance (first bin b with
// sum_g cum[g][b + 1] >= k, from the UNclamped histograms); shard g had to send cum[g][T + 1] entries.  The largest such
// count over all (shard, query) pairs of the launch goes to need_out: the exchange was exact iff it is <= the prefix sent
// (the caller clamps it to min(k, shard rows) first: with many ties at the k-th distance the raw count exceeds what a
// shard can owe).  cum_ld: row pitch of the histograms in uint32.
__device__ __forceinline__ void merge_report_need(const uint32_t *__restrict__ cum, int64_t cum_ld, int G, int Q, int qi, int nbins, int k,
                                                  int32_t *need_out, int lane)
{
    int T = nbins - 1;
    for (int c = 2; c >= 0; --c) {
        const int b = 64 * c + lane;
        uint32_t sb = 0;
        if (b < nbins)
            for (int g = 0; g < G; ++g) sb += cum[((int64_t)g * Q + qi) * cum_ld + b + 1];
        const uint64_t m = __ballot(b < nbins && sb >= (uint32_t)k);
        if (m) T = 64 * c + __builtin_ctzll(m);
    }
    uint32_t nd = 0;
    for (int g = lane; g < G; g += 64) nd = max(nd, cum[((int64_t)g * Q + qi) * cum_ld + T + 1]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) nd = max(nd, (uint32_t)__shfl_xor((int)nd, d, 64));
    if (lane == 0) atomicMax(need_out, (int32_t)min(nd, 0x7fffffffu));
}

// Compact form of the same merge for the sharded search: a shard does not send its distance rows (1 byte per
// entry) but its cumulative distance histogram (cum[b] = rows of the shard with distance < b, nbins + 1 words per
// query -- the k_hamming_topk by-product), and its list as 16-bit LOCAL row numbers (shards of <= 65536 rows).
// A sorted list is fully described by its histogram: run boundaries are min(cum[b], kin) directly, an entry's
// distance is the bin its position falls in (binary search over the boundaries in LDS), its global index is
// local + g * shard_rows.  Same output as k_merge_sorted on the expanded inputs.
__global__ __launch_bounds__(256) void k_merge_cum(const uint16_t *__restrict__ idx_local,
                                                   const uint32_t *__restrict__ cum, int G, int Q, int kin,
                                                   int64_t shard_rows, int32_t *__restrict__ idx_out,
                                                   uint8_t *__restrict__ dist_out, int k, int nbins,
                                                   int32_t *__restrict__ need_out)
{
    extern __shared__ uint4 lds4[];
    int32_t *start = reinterpret_cast<int32_t *>(lds4);           // [G][nbins + 1]: first position with dist >= b
    int32_t *delta = start + G * (nbins + 1);                     // [G][nbins]
    uint32_t *base = reinterpret_cast<uint32_t *>(delta + G * nbins);   // [nbins + 1]
    const int qi = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    for (int u = tid; u < G * (nbins + 1); u += 256) {
        const int g = u / (nbins + 1), b = u - g * (nbins + 1);
        start[u] = (int32_t)min(cum[((int64_t)g * Q + qi) * (nbins + 1) + b], (uint32_t)kin);
    }
    __syncthreads();
    if (need_out && wv == 1) merge_report_need(cum, nbins + 1, G, Q, qi, nbins, k, need_out, lane);
    if (wv == 0) {   // totals per bin and their exclusive scan (up to 3 bins per lane)
        uint32_t t[3] = {0, 0, 0};
        const int b0 = 3 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (b0 + j < nbins)
                for (int g = 0; g < G; ++g) t[j] += (uint32_t)(start[g * (nbins + 1) + b0 + j + 1] - start[g * (nbins + 1) + b0 + j]);
        const uint32_t incl = wave_incl_scan_u32(t[0] + t[1] + t[2]);
        const uint32_t excl = incl - (t[0] + t[1] + t[2]);
        if (b0 < nbins) base[b0] = excl;
        if (b0 + 1 < nbins) base[b0 + 1] = excl + t[0];
        if (b0 + 2 < nbins) base[b0 + 2] = excl + t[0] + t[1];
    }
    __syncthreads();
    for (int b = tid; b < nbins; b += 256) {
        uint32_t run = base[b];
        for (int g = 0; g < G; ++g) {
            const int s0 = start[g * (nbins + 1) + b], s1 = start[g * (nbins + 1) + b + 1];
            delta[g * nbins + b] = (int32_t)run - s0;
            run += (uint32_t)(s1 - s0);
        }
    }
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const uint16_t *irow = idx_local + ((int64_t)g * Q + qi) * kin;
        const int32_t *st = start + g * (nbins + 1);
        const int len = st[nbins];                                // entries this shard really sent
        for (int p = tid; p < len; p += 256) {
            int lo = 0, hi = nbins;                               // invariant: st[lo] <= p < st[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (st[mid] <= p) lo = mid;
                else hi = mid;
            }
            const int pos = p + delta[g * nbins + lo];
            if (pos < k) {
                idx_out[(int64_t)qi * k + pos] = (int32_t)((int64_t)irow[p] + g * shard_rows);
                if (dist_out) dist_out[(int64_t)qi * k + pos] = (uint8_t)lo;
            }
        }
    }
}

static int set_lds_attr(const void *fn, size_t bytes, const char *what)
{
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
    }
    return WV_OK;
}

// calls f(std::bool_constant<U16>, std::bool_constant<STAGED>) for the runtime pair
template <typename F>
static int dispatch_rank(bool u16, bool staged, F f)
{
    if (u16) return staged ? f(std::true_type{}, std::true_type{}) : f(std::true_type{}, std::false_type{});
    return staged ? f(std::false_type{}, std::true_type{}) : f(std::false_type{}, std::false_type{});
}

template <int WORDS>
static int launch_transpose(const uint64_t *db, uint64_t *dbT, int64_t N, hipStream_t st)
{
    const int C = (int)ceil_div(N, kTopkThreads);
    const int64_t total = (int64_t)C * kTopkThreads;
    hipLaunchKernelGGL((k_transpose_db<WORDS>), dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096)),
                       dim3(256), 0, st, db, dbT, N, C);
    WV_CHECK_LAUNCH("k_transpose_db");
    return WV_OK;
}

// dbT == nullptr: build the column image into `ws` first (one extra small launch per call)
template <int WORDS>
static int launch_topk(const uint64_t *q, const uint64_t *db, const uint64_t *dbT_ready, int32_t *idx, uint8_t *dist,
                       int Q, int64_t N, int nbits, int k, int64_t idx_offset, void *ws, hipStream_t st,
                       uint32_t *cum = nullptr)
{
    const int C = (int)ceil_div(N, kTopkThreads);
    const int nbins = nbits + 1;
    const uint64_t *dbT = dbT_ready;
    if (const int tpq = rank2_tpq(Q, N, k)) {
        const void *img = nullptr;
        if (dbT_ready) {
            img = (const char *)dbT_ready + (tpq == 256 ? r2_off256(N, WORDS) : r2_off64(N, WORDS));
        } else {
            int rc0 = rank2_prepare(db, ws, N, WORDS, tpq, st);
            if (rc0) return rc0;
            img = ws;
        }
        const int rc2 = rank2_launch(q, img, idx, nullptr, dist, Q, N, nbits, k, idx_offset, cum, tpq, st);
        if (rc2 <= 0) return rc2;                                // 1 = shape not covered after all: first-generation kernel
    }
    if (!dbT) {
        int rc0 = launch_transpose<WORDS>(db, (uint64_t *)ws, N, st);
        if (rc0) return rc0;
        dbT = (const uint64_t *)ws;
    }
    const bool u16 = rank_u16(N), staged = rank_staged(k, nbins, u16);
    const size_t lds = rank_lds_bytes(nbins, u16, k);
    if (const char *e = ::wv::tune("WV_TOPK_DBG")) {
        const int v = atoi(e);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_topk_dbg), &v, sizeof(int));
    }
    int rc = dispatch_rank(u16, staged, [&](auto U, auto S) {
        auto kern = k_hamming_topk<WORDS, decltype(U)::value, decltype(S)::value>;
        int r0 = set_lds_attr(reinterpret_cast<const void *>(kern), lds, "hamming_topk");
        if (r0) return r0;
        hipLaunchKernelGGL(kern, dim3(Q), dim3(kTopkThreads), lds, st, q, dbT, idx, dist, N, C, nbins, k, idx_offset, cum);
        return (int)WV_OK;
    });
    if (rc) return rc;
    WV_CHECK_LAUNCH("k_hamming_topk");
    return WV_OK;
}

size_t topk_prepared_bytes(int64_t N, int words)
{
    size_t b = r2_off64(N, words);
    if (N <= kImg64MaxRows) b += rank2_image_bytes(N, words, 64);
    return b;
}

int topk_prepare(const uint64_t *db, void *dbT, int64_t N, int words, hipStream_t st)
{
    int rc = words == 1 ? launch_transpose<1>(db, (uint64_t *)dbT, N, st) : launch_transpose<2>(db, (uint64_t *)dbT, N, st);
    if (!rc && N <= kImg256MaxRows) rc = rank2_prepare(db, (char *)dbT + r2_off256(N, words), N, words, 256, st);
    if (!rc && N <= kImg64MaxRows) rc = rank2_prepare(db, (char *)dbT + r2_off64(N, words), N, words, 64, st);
    return rc;
}

}  // namespace wv

using namespace wv;

extern "C" size_t wv_hamming_topk_workspace_bytes(int Q, int64_t N, int words, int k)
{
    (void)Q; (void)k;
    if (N <= 0 || words <= 0) return 0;
    // one database image, for whichever kernel the call takes
    return std::max(std::max(old_image_bytes(N, words), rank2_image_bytes(N, words, 256)), rank2_image_bytes(N, words, 64));
}

extern "C" int wv_hamming_topk(const uint64_t *q, const uint64_t *db, int32_t *idx, uint8_t *dist,
                               int Q, int64_t N, int nbits, int k, int64_t idx_offset,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(q && db && idx, "hamming_topk: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_topk: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_topk: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_topk: k=%d must be in [1, N=%lld] (torch.topk raises too)", k,
               (long long)N);
    WV_REQUIRE(N + idx_offset <= 0x7fffffffLL && idx_offset >= 0, "hamming_topk: indices exceed int32");
    const int words = (nbits + 63) / 64;
    const size_t need = wv_hamming_topk_workspace_bytes(Q, N, words, k);
    if (!workspace || workspace_bytes < need)
        WV_FAIL(WV_ENOMEM, "hamming_topk: workspace %zu < %zu bytes", workspace_bytes, need);
    if (Q == 0) return WV_OK;
    hipStream_t st = (hipStream_t)stream;
    if (words == 1) return launch_topk<1>(q, db, nullptr, idx, dist, Q, N, nbits, k, idx_offset, workspace, st);
    return launch_topk<2>(q, db, nullptr, idx, dist, Q, N, nbits, k, idx_offset, workspace, st);
}

extern "C" size_t wv_db_prepared_bytes(int64_t N, int words)
{
    if (N <= 0 || words < 1 || words > 4) return 0;
    return align_up((int64_t)dist_prepared_bytes(N, words), 256) + (words <= 2 ? topk_prepared_bytes(N, words) : 0);
}

extern "C" int wv_db_prepare(const uint64_t *db, int64_t N, int words, void *prepared, size_t prepared_bytes,
                             void *stream)
{
    WV_REQUIRE(db && prepared, "db_prepare: null buffer");
    WV_REQUIRE(N >= 1 && words >= 1 && words <= 4, "db_prepare: bad shape N=%lld words=%d", (long long)N, words);
    const size_t need = wv_db_prepared_bytes(N, words);
    if (prepared_bytes < need) WV_FAIL(WV_ENOMEM, "db_prepare: buffer %zu < %zu bytes", prepared_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    int rc = dist_prepare(db, prepared, N, words, st);
    if (rc) return rc;
    if (words <= 2)
        rc = topk_prepare(db, (char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256), N, words, st);
    return rc;
}

extern "C" int wv_hamming_topk_prepared(const uint64_t *q, const void *prepared, int32_t *idx, uint8_t *dist, int Q,
                                        int64_t N, int nbits, int k, int64_t idx_offset, void *stream)
{
    WV_REQUIRE(q && prepared && idx, "hamming_topk_prepared: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_topk_prepared: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_topk_prepared: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_topk_prepared: k=%d must be in [1, N=%lld]", k, (long long)N);
    WV_REQUIRE(N + idx_offset <= 0x7fffffffLL && idx_offset >= 0, "hamming_topk_prepared: indices exceed int32");
    if (Q == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    const uint64_t *dbT = (const uint64_t *)((const char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256));
    hipStream_t st = (hipStream_t)stream;
    if (words == 1) return launch_topk<1>(q, nullptr, dbT, idx, dist, Q, N, nbits, k, idx_offset, nullptr, st);
    return launch_topk<2>(q, nullptr, dbT, idx, dist, Q, N, nbits, k, idx_offset, nullptr, st);
}

extern "C" int wv_topk_merge(const int32_t *idx_in, const uint8_t *dist_in, int G, int Q, int kin,
                             int32_t *idx_out, uint8_t *dist_out, int k, int nbits, void *stream)
{
    WV_REQUIRE(idx_in && dist_in && idx_out, "topk_merge: null buffer");
    WV_REQUIRE(G >= 1 && Q >= 0 && kin >= 1, "topk_merge: bad shape G=%d Q=%d kin=%d", G, Q, kin);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "topk_merge: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && (int64_t)k <= (int64_t)G * kin, "topk_merge: k=%d > G*kin", k);
    if (Q == 0) return WV_OK;
    const int nbins = nbits + 2;  // dist = nbits + 1 marks padding entries: they rank after every real one
    const size_t lds = ((size_t)G * (nbins + 1) + (size_t)G * nbins + nbins + 1 + 4) * 4;
    WV_REQUIRE(lds <= 60 * 1024, "topk_merge: too many shards (G=%d)", G);
    hipLaunchKernelGGL(k_merge_sorted, dim3(Q), dim3(256), lds, (hipStream_t)stream, idx_in, dist_in, G, Q, kin,
                       idx_out, dist_out, k, nbins);
    WV_CHECK_LAUNCH("k_merge_sorted");
    return WV_OK;
}

namespace wv {
// Sharded mAP, receiving side: per query the G shards' relevance strings (bit p = is the shard's p-th nearest row relevant)
// and cumulative histograms -> average precision.  The global list orders entries by (distance, shard, position), so the
// merged relevance string is the concatenation, bin by bin and shard by shard, of runs of the shards' strings; it is
// assembled in LDS (k bits) with shifted 32-bit copies and walked exactly as k_map_at_k walks a list (ap_walk.hpp).
// need_out: as in k_merge_cum.
__global__ __launch_bounds__(256) void k_merge_relbits_ap(const uint32_t *__restrict__ relbits, const uint32_t *__restrict__ cum,
                                                          int G, int Q, int kin, int w32, int k, int nbins,
                                                          float *__restrict__ ap, int32_t *__restrict__ nrel,
                                                          int32_t *__restrict__ need_out, int64_t rb_ld32, int64_t cum_ld)
{
    extern __shared__ uint4 lds4[];
    int32_t *start = reinterpret_cast<int32_t *>(lds4);           // [G][nbins + 1]: first position with dist >= b
    uint32_t *base = reinterpret_cast<uint32_t *>(start + G * (nbins + 1));   // [nbins + 1]
    uint32_t *M = base + nbins + 1;                               // merged relevance string, k bits (+ spill word)
    const int mwords = (k + 31) / 32 + 1;
    uint32_t *scratch = M + mwords + (mwords & 1);                // ap_finish: 8-byte aligned
    const int qi = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    for (int u = tid; u < G * (nbins + 1); u += 256) {
        const int g = u / (nbins + 1), b = u - g * (nbins + 1);
        start[u] = (int32_t)min(cum[((int64_t)g * Q + qi) * cum_ld + b], (uint32_t)kin);
    }
    for (int u = tid; u < mwords; u += 256) M[u] = 0;
    __syncthreads();
    if (need_out && wv == 1) merge_report_need(cum, cum_ld, G, Q, qi, nbins, k, need_out, lane);
    if (wv == 0) {   // totals per bin and their exclusive scan (up to 3 bins per lane)
        uint32_t t[3] = {0, 0, 0};
        const int b0 = 3 * lane;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (b0 + j < nbins)
                for (int g = 0; g < G; ++g) t[j] += (uint32_t)(start[g * (nbins + 1) + b0 + j + 1] - start[g * (nbins + 1) + b0 + j]);
        const uint32_t incl = wave_incl_scan_u32(t[0] + t[1] + t[2]);
        const uint32_t excl = incl - (t[0] + t[1] + t[2]);
        if (b0 < nbins) base[b0] = excl;
        if (b0 + 1 < nbins) base[b0 + 1] = excl + t[0];
        if (b0 + 2 < nbins) base[b0 + 2] = excl + t[0] + t[1];
    }
    __syncthreads();
    for (int u = tid; u < G * nbins; u += 256) {                  // one run per (bin, shard)
        const int b = u / G, g = u - b * G;
        const int s0 = start[g * (nbins + 1) + b], n0 = start[g * (nbins + 1) + b + 1] - s0;
        if (n0 <= 0) continue;
        uint32_t dst = base[b];
        for (int g2 = 0; g2 < g; ++g2) dst += (uint32_t)(start[g2 * (nbins + 1) + b + 1] - start[g2 * (nbins + 1) + b]);
        if (dst >= (uint32_t)k) continue;
        const int n = min(n0, k - (int)dst);
        const uint32_t *src = relbits + ((int64_t)g * Q + qi) * rb_ld32;
        for (int o = 0; o < n; o += 32) {
            const int cnt = min(32, n - o), sp = s0 + o, sw = sp >> 5, sh = sp & 31;
            uint32_t v = src[sw] >> sh;
            if (sh && sw + 1 < w32) v |= src[sw + 1] << (32 - sh);
            if (cnt < 32) v &= (1u << cnt) - 1u;
            if (!v) continue;
            const uint32_t dp = dst + (uint32_t)o, dw = dp >> 5, ds = dp & 31;
            atomicOr(&M[dw], v << ds);
            if (ds && (v >> (32 - ds))) atomicOr(&M[dw + 1], v >> (32 - ds));
        }
    }
    __syncthreads();
    const int R = (k + 255) / 256;
    double *wsum = reinterpret_cast<double *>(scratch + kApRounds * 4 + (kApRounds * 4 & 1));
    ApState st;
    for (int c0 = 0; c0 < R; c0 += kApRounds) {                   // chunks of 32 rounds: any k (mAP@ALL: k = database size)
        const int Rc = min(kApRounds, R - c0);
        uint32_t mine = 0;
        for (int r = 0; r < Rc; ++r) {
            const int p = (c0 + r) * 256 + tid;
            const bool rel = p < k && ((M[p >> 5] >> (p & 31)) & 1u);
            mine |= (rel ? 1u : 0u) << r;
            const uint64_t m = __ballot(rel);
            if (lane == 0) scratch[r * 4 + wv] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        ap_accum<256>(mine, scratch, Rc, c0, tid, st);
        __syncthreads();
    }
    ap_final<256>(st, wsum, tid, ap + qi, nrel ? nrel + qi : nullptr, [] { __syncthreads(); });
}
}  // namespace wv

extern "C" int wv_topk_merge_cum_need(const uint16_t *idx_local, const uint32_t *cum, int G, int Q, int kin, int64_t shard_rows,
                                      int32_t *idx_out, uint8_t *dist_out, int k, int nbits, int32_t *need_out, void *stream);

extern "C" int wv_topk_merge_cum(const uint16_t *idx_local, const uint32_t *cum, int G, int Q, int kin,
                                 int64_t shard_rows, int32_t *idx_out, uint8_t *dist_out, int k, int nbits, void *stream)
{
    return wv_topk_merge_cum_need(idx_local, cum, G, Q, kin, shard_rows, idx_out, dist_out, k, nbits, nullptr, stream);
}

extern "C" int wv_topk_merge_cum_need(const uint16_t *idx_local, const uint32_t *cum, int G, int Q, int kin, int64_t shard_rows,
                                      int32_t *idx_out, uint8_t *dist_out, int k, int nbits, int32_t *need_out, void *stream)
{
    WV_REQUIRE(idx_local && cum && idx_out, "topk_merge_cum: null buffer");
    WV_REQUIRE(G >= 1 && Q >= 0 && kin >= 1 && k >= 1, "topk_merge_cum: bad shape G=%d Q=%d kin=%d k=%d", G, Q, kin, k);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "topk_merge_cum: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(shard_rows >= 1 && shard_rows <= 65536, "topk_merge_cum: %lld rows per shard do not fit 16-bit local indices",
               (long long)shard_rows);
    WV_REQUIRE((int64_t)G * shard_rows <= 0x7fffffffLL, "topk_merge_cum: indices exceed int32");
    if (Q == 0) return WV_OK;
    const int nbins = nbits + 1;
    const size_t lds = ((size_t)G * (nbins + 1) + (size_t)G * nbins + nbins + 1 + 4) * 4;
    WV_REQUIRE(lds <= 60 * 1024, "topk_merge_cum: too many shards (G=%d)", G);
    hipLaunchKernelGGL(k_merge_cum, dim3(Q), dim3(256), lds, (hipStream_t)stream, idx_local, cum, G, Q, kin, shard_rows,
                       idx_out, dist_out, k, nbins, need_out);
    WV_CHECK_LAUNCH("k_merge_cum");
    return WV_OK;
}

extern "C" int wv_rank_from_dist(const uint8_t *dist_matrix, int64_t ld_dist, int Q, int64_t N,
                                 int nbits, int32_t *idx, uint8_t *dist, int k, void *stream)
{
    WV_REQUIRE(dist_matrix && idx, "rank_from_dist: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1 && ld_dist >= N, "rank_from_dist: bad shape");
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "rank_from_dist: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N && N <= 0x7fffffffLL, "rank_from_dist: k=%d must be in [1, N]", k);
    if (Q == 0) return WV_OK;
    const int C = (int)ceil_div(N, kTopkThreads);
    const int nbins = nbits + 1;
    const bool u16 = rank_u16(N), staged = rank_staged(k, nbins, u16);
    const size_t lds = rank_lds_bytes(nbins, u16, k);
    int rc = dispatch_rank(u16, staged, [&](auto U, auto S) {
        auto kern = k_rank_from_dist<decltype(U)::value, decltype(S)::value>;
        int r0 = set_lds_attr(reinterpret_cast<const void *>(kern), lds, "rank_from_dist");
        if (r0) return r0;
        hipLaunchKernelGGL(kern, dim3(Q), dim3(kTopkThreads), lds, (hipStream_t)stream, dist_matrix, ld_dist, N, idx,
                           dist, k, C, nbins);
        return (int)WV_OK;
    });
    if (rc) return rc;
    WV_CHECK_LAUNCH("k_rank_from_dist");
    return WV_OK;
}

extern "C" int wv_map_at_k_ld(const int32_t *idx, int64_t ld, int Q, int k, const uint64_t *qlab,
                              const uint64_t *dblab, int lwords, float *ap, int32_t *nrel, void *stream)
{
    WV_REQUIRE(idx && qlab && dblab && ap, "map_at_k: null buffer");
    WV_REQUIRE(Q >= 0 && k >= 1 && lwords >= 1 && ld >= k, "map_at_k: bad shape Q=%d k=%d ld=%lld lwords=%d", Q, k, (long long)ld, lwords);
    if (Q == 0) return WV_OK;
    hipStream_t st = (hipStream_t)stream;
    if (lwords == 1)
        hipLaunchKernelGGL((k_map_at_k<1>), dim3(Q), dim3(256), 0, st, idx, ld, k, qlab, dblab, lwords, ap, nrel);
    else if (lwords == 2)
        hipLaunchKernelGGL((k_map_at_k<2>), dim3(Q), dim3(256), 0, st, idx, ld, k, qlab, dblab, lwords, ap, nrel);
    else
        hipLaunchKernelGGL((k_map_at_k<0>), dim3(Q), dim3(256), 0, st, idx, ld, k, qlab, dblab, lwords, ap, nrel);
    WV_CHECK_LAUNCH("k_map_at_k");
    return WV_OK;
}

extern "C" int wv_map_at_k(const int32_t *idx, int Q, int k, const uint64_t *qlab,
                           const uint64_t *dblab, int lwords, float *ap, int32_t *nrel, void *stream)
{
    return wv_map_at_k_ld(idx, k, Q, k, qlab, dblab, lwords, ap, nrel, stream);
}

extern "C" int wv_hit_prefix(const int32_t *idx, int Q, int k, const uint64_t *qlab, const uint64_t *dblab,
                             int lwords, uint32_t *hits, void *stream)
{
    WV_REQUIRE(idx && qlab && dblab && hits, "hit_prefix: null buffer");
    WV_REQUIRE(Q >= 0 && k >= 1 && lwords >= 1, "hit_prefix: bad shape Q=%d k=%d lwords=%d", Q, k, lwords);
    if (Q == 0) return WV_OK;
    hipLaunchKernelGGL(k_hit_prefix, dim3(Q), dim3(256), 0, (hipStream_t)stream, idx, k, qlab, dblab, lwords, hits);
    WV_CHECK_LAUNCH("k_hit_prefix");
    return WV_OK;
}

extern "C" int wv_hamming_topk_ex(const uint64_t *q, const uint64_t *db, const void *prepared, int32_t *idx,
                                  uint8_t *dist, uint32_t *cum, int Q, int64_t N, int nbits, int k, int64_t idx_offset,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(q && idx && (db || prepared), "hamming_topk_ex: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_topk_ex: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_topk_ex: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_topk_ex: k=%d must be in [1, N=%lld]", k, (long long)N);
    WV_REQUIRE(N + idx_offset <= 0x7fffffffLL && idx_offset >= 0, "hamming_topk_ex: indices exceed int32");
    if (Q == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t *dbT = nullptr;
    if (prepared) {
        dbT = (const uint64_t *)((const char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256));
    } else {
        const size_t need = wv_hamming_topk_workspace_bytes(Q, N, words, k);
        if (!workspace || workspace_bytes < need)
            WV_FAIL(WV_ENOMEM, "hamming_topk_ex: workspace %zu < %zu bytes", workspace_bytes, need);
    }
    if (words == 1) return launch_topk<1>(q, db, dbT, idx, dist, Q, N, nbits, k, idx_offset, workspace, st, cum);
    return launch_topk<2>(q, db, dbT, idx, dist, Q, N, nbits, k, idx_offset, workspace, st, cum);
}

// ---------------------------------------------------------------------------------- mAP without the lists
extern "C" size_t wv_rank_labels_prepared_bytes(int64_t N, int lwords)
{
    if (N < 1 || N > kImg256MaxRows || lwords < 1 || lwords > 2) return 0;
    return rank2_labels_bytes(N, lwords);
}

extern "C" int wv_rank_labels_prepare(const uint64_t *dblab, int64_t N, int lwords, void *prepared_labels, size_t prepared_bytes,
                                      void *stream)
{
    WV_REQUIRE(dblab && prepared_labels, "rank_labels_prepare: null buffer");
    const size_t need = wv_rank_labels_prepared_bytes(N, lwords);
    if (!need)
        WV_FAIL(WV_ENOTSUP, "rank_labels_prepare: %lld rows x %d label words are outside the fused kernels (<= 32768 rows, <= 128 classes)",
                (long long)N, lwords);
    if (prepared_bytes < need) WV_FAIL(WV_ENOMEM, "rank_labels_prepare: buffer %zu < %zu bytes", prepared_bytes, need);
    return rank2_labels_prepare(dblab, prepared_labels, N, lwords, (hipStream_t)stream);
}

extern "C" int wv_hamming_map_at_k(const uint64_t *q, const void *prepared, const void *prepared_labels, const uint64_t *qlab,
                                   int lwords, int Q, int64_t N, int nbits, int k, float *ap, int32_t *nrel, void *stream)
{
    WV_REQUIRE(lwords >= 1, "hamming_map_at_k: lwords=%d", lwords);
    if (lwords > 2) WV_FAIL(WV_ENOTSUP, "hamming_map_at_k: %d label words (more than 128 classes): wv_hamming_topk + wv_map_at_k", lwords);
    WV_REQUIRE(q && prepared && prepared_labels && qlab && ap, "hamming_map_at_k: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_map_at_k: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_map_at_k: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_map_at_k: k=%d must be in [1, N=%lld]", k, (long long)N);
    if (Q == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    int tpq = rank2_tpq(Q, N, k);
    if (tpq == 64 && k > 32 * 64 && N <= kImg256MaxRows) tpq = 256;      // the AP walk keeps 32 list positions per thread
    if (!tpq || N > kImg256MaxRows || (tpq == 64 && N > kImg64MaxRows))
        WV_FAIL(WV_ENOTSUP, "hamming_map_at_k: %lld rows / k=%d are outside the windowed kernel (wv_hamming_topk + wv_map_at_k)",
                (long long)N, k);
    const char *base = (const char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256);
    const void *img = base + (tpq == 256 ? r2_off256(N, words) : r2_off64(N, words));
    const int rc = rank2_launch(q, img, nullptr, nullptr, nullptr, Q, N, nbits, k, 0, nullptr, tpq, (hipStream_t)stream,
                                prepared_labels, qlab, ap, nrel, nullptr, 0, 0, lwords);
    if (rc > 0) WV_FAIL(WV_ENOTSUP, "hamming_map_at_k: k=%d is outside the fused kernel (wv_hamming_topk + wv_map_at_k)", k);
    return rc;
}

// ---------------------------------------------------------------------------------- sharded search, two steps
static int shard_call(const char *what, const uint64_t *q, const uint64_t *db, const void *prepared, uint16_t *rows16,
                      uint32_t *cum, int Q, int64_t N, int nbits, int k, void *workspace, size_t workspace_bytes, void *stream)
{
    const int words = (nbits + 63) / 64;
    const int tpq = rank2_tpq(Q, N, std::max(k, 1));
    if (!tpq) WV_FAIL(WV_ENOTSUP, "%s: shards of %lld rows / lists of %d entries are outside the windowed kernel "
                                  "(rows <= 32768, 16-bit row numbers)", what, (long long)N, k);
    hipStream_t st = (hipStream_t)stream;
    const void *img = nullptr;
    if (prepared) {
        const char *base = (const char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256);
        img = base + (tpq == 256 ? r2_off256(N, words) : r2_off64(N, words));
    } else {
        const size_t need = wv_hamming_topk_workspace_bytes(Q, N, words, k);
        if (!workspace || workspace_bytes < need) WV_FAIL(WV_ENOMEM, "%s: workspace %zu < %zu bytes", what, workspace_bytes, need);
        int rc0 = rank2_prepare(db, workspace, N, words, tpq, st);
        if (rc0) return rc0;
        img = workspace;
    }
    const int rc = rank2_launch(q, img, nullptr, rows16, nullptr, Q, N, nbits, k, 0, cum, tpq, st);
    if (rc > 0) WV_FAIL(WV_ENOTSUP, "%s: shape outside the windowed kernel", what);
    return rc;
}

extern "C" int wv_hamming_hist(const uint64_t *q, const uint64_t *db, const void *prepared, uint32_t *cum, int Q, int64_t N,
                               int nbits, void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(q && cum && (db || prepared), "hamming_hist: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_hist: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_hist: nbits=%d (supported: 1..128)", nbits);
    if (Q == 0) return WV_OK;
    return shard_call("hamming_hist", q, db, prepared, nullptr, cum, Q, N, nbits, 0, workspace, workspace_bytes, stream);
}

extern "C" int wv_hamming_shard_prefix(const uint64_t *q, const uint64_t *db, const void *prepared, uint16_t *rows, uint32_t *cum,
                                       int Q, int64_t N, int nbits, int k, void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(q && rows && cum && (db || prepared), "hamming_shard_prefix: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_shard_prefix: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_shard_prefix: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_shard_prefix: k=%d must be in [1, N=%lld]", k, (long long)N);
    if (Q == 0) return WV_OK;
    return shard_call("hamming_shard_prefix", q, db, prepared, rows, cum, Q, N, nbits, k, workspace, workspace_bytes, stream);
}

extern "C" int wv_hamming_shard_relbits(const uint64_t *q, const void *prepared, const void *prepared_labels, const uint64_t *qlab,
                                        int lwords, uint64_t *relbits, int64_t relbits_ld, uint32_t *cum, int64_t cum_ld, int Q,
                                        int64_t N, int nbits, int k, void *stream)
{
    WV_REQUIRE(lwords >= 1, "hamming_shard_relbits: lwords=%d", lwords);
    if (lwords > 2) WV_FAIL(WV_ENOTSUP, "hamming_shard_relbits: %d label words (more than 128 classes)", lwords);
    WV_REQUIRE((relbits_ld == 0 || relbits_ld >= (k + 63) / 64) && (cum_ld == 0 || cum_ld >= nbits + 2),
               "hamming_shard_relbits: row pitches %lld / %lld too small", (long long)relbits_ld, (long long)cum_ld);
    WV_REQUIRE(q && prepared && prepared_labels && qlab && relbits && cum, "hamming_shard_relbits: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_shard_relbits: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_shard_relbits: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_shard_relbits: k=%d must be in [1, N=%lld]", k, (long long)N);
    if (Q == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    int tpq = rank2_tpq(Q, N, k);
    if (tpq == 64 && k > 32 * 64 && N <= kImg256MaxRows) tpq = 256;
    if (!tpq || N > kImg256MaxRows || (tpq == 64 && N > kImg64MaxRows))
        WV_FAIL(WV_ENOTSUP, "hamming_shard_relbits: %lld rows / k=%d are outside the windowed kernel", (long long)N, k);
    const char *base = (const char *)prepared + align_up((int64_t)dist_prepared_bytes(N, words), 256);
    const void *img = base + (tpq == 256 ? r2_off256(N, words) : r2_off64(N, words));
    const int rc = rank2_launch(q, img, nullptr, nullptr, nullptr, Q, N, nbits, k, 0, cum, tpq, (hipStream_t)stream, prepared_labels,
                                qlab, nullptr, nullptr, relbits, relbits_ld, cum_ld, lwords);
    if (rc > 0) WV_FAIL(WV_ENOTSUP, "hamming_shard_relbits: k=%d is outside the fused kernel", k);
    return rc;
}

extern "C" int wv_merge_relbits_map(const uint64_t *relbits, int64_t relbits_ld, const uint32_t *cum, int64_t cum_ld, int G, int Q,
                                    int kin, int k, int nbits, float *ap, int32_t *nrel, int32_t *need_out, void *stream)
{
    WV_REQUIRE((relbits_ld == 0 || relbits_ld >= (kin + 63) / 64) && (cum_ld == 0 || cum_ld >= nbits + 2),
               "merge_relbits_map: row pitches %lld / %lld too small", (long long)relbits_ld, (long long)cum_ld);
    WV_REQUIRE(relbits && cum && ap, "merge_relbits_map: null buffer");
    WV_REQUIRE(G >= 1 && Q >= 0 && kin >= 1 && k >= 1, "merge_relbits_map: bad shape G=%d Q=%d kin=%d k=%d", G, Q, kin, k);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "merge_relbits_map: nbits=%d (supported: 1..128)", nbits);
    if (Q == 0) return WV_OK;
    const int nbins = nbits + 1, w32 = 2 * (int)ceil_div(kin, 64), mwords = (k + 31) / 32 + 1;
    const size_t lds = ((size_t)G * (nbins + 1) + nbins + 1 + mwords + (mwords & 1) + ap_scratch_dwords<256>() + 4) * 4;
    WV_REQUIRE(lds <= 60 * 1024, "merge_relbits_map: too many shards (G=%d)", G);
    hipLaunchKernelGGL(k_merge_relbits_ap, dim3(Q), dim3(256), lds, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(relbits),
                       cum, G, Q, kin, w32, k, nbins, ap, nrel, need_out, relbits_ld ? 2 * relbits_ld : (int64_t)w32,
                       cum_ld ? cum_ld : (int64_t)(nbins + 1));
    WV_CHECK_LAUNCH("k_merge_relbits_ap");
    return WV_OK;
}

extern "C" int wv_hamming_topk_rows16(const uint64_t *q, const uint64_t *db, const void *prepared, uint16_t *rows, int Q,
                                      int64_t N, int nbits, int k, void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(q && rows && (db || prepared), "hamming_topk_rows16: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "hamming_topk_rows16: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_topk_rows16: nbits=%d (supported: 1..128)", nbits);
    WV_REQUIRE(k >= 1 && k <= N, "hamming_topk_rows16: k=%d must be in [1, N=%lld]", k, (long long)N);
    if (Q == 0) return WV_OK;
    return shard_call("hamming_topk_rows16", q, db, prepared, rows, nullptr, Q, N, nbits, k, workspace, workspace_bytes, stream);
}

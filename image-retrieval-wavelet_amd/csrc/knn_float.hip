// Real-valued k-NN for gfx950: fp32 score matrix on the matrix cores + exact stable ranking.
//
// Reference: get_knn_torch (/root/reference/main/engine/get_knn.py:60-71)
//   hamming / cosine : scores = q @ r.T ; torch.topk(largest=True)
//   l2               : torch.cdist(q, r, p=2) ; torch.topk(largest=False)
//     (for more than 25 rows cdist itself evaluates sqrt(max(0, |q|^2 + |r|^2 - 2 q.r)) through a
//      matmul, which is the form used here)
//
// k_scores: one wave owns a 64x64 tile of the [Q, N] score matrix as 2x2 blocks of
//   v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate: bit-for-bit an fmaf chain, so no precision
//   is traded for the matrix cores).  Lane l feeds row (l & 31) of A and of B; lane half h = l>>5
//   covers k in [8c+4h, 8c+4h+4) of every 8-wide chunk with ONE float4 per operand -- the MFMA
//   sums over k, so any k <-> (step, half) bijection shared by A and B is valid.  Both operands are
//   read from fragment images (k_pack_fragments) in which those 64 float4 are contiguous.
// k_radix_pass: exact ranking of every row by 6 stable LSD counting-sort passes over the
//   order-preserving 32-bit image of the float (ties therefore end in ascending index order),
//   with the same private-column LDS histogram as the Hamming ranking kernel (topk.hip).
#include "common.hpp"

namespace wv {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kRadixBits = 6;
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kRadixThreads = 256;

__global__ __launch_bounds__(256) void k_row_sqnorm(const float *__restrict__ x, int64_t rows, int D,
                                                    float *__restrict__ out)
{
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float v = x[row * D + k];
        s = fmaf(v, v, s);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) out[row] = s;
}

// Fragment image of an embedding matrix x [rows, D]: for every block of 32 rows and every chunk of 8 k-values the 64
// float4 a wave feeds to four v_mfma_f32_32x32x2_f32 -- lane (r, h) holds x[32 rb + r][8c + 4h .. 8c + 4h + 4) -- lie in
// lane order, 1 KB contiguous.  k_scores then reads each operand fragment with ONE fully coalesced load instruction
// (the row-major matrix gave 32 B per row per instruction, a quarter of every line it pulled through L1: the operand
// stream, not the matrix pipe, set its pace).  Rows beyond `rows` and k beyond D are zeros; a workgroup writes the four
// chunks that share one 128-byte line of each source row.
__global__ __launch_bounds__(256) void k_pack_fragments(const float *__restrict__ x, float4 *__restrict__ out,
                                                        int64_t rows, int D, int nchunk, int64_t total)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index of the image
    if (o >= total) return;
    const int lane = (int)(o & 63);
    const int64_t bc = o >> 6;
    const int c = (int)(bc % nchunk);
    const int64_t row = (bc / nchunk) * 32 + (lane & 31);
    const int k = 8 * c + 4 * (lane >> 5);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows && k < D) {
        const float *src = x + row * D + k;
        if ((D & 3) == 0) {              // rows are 16-byte aligned: one load
            v = *reinterpret_cast<const float4 *>(src);
        } else {                         // any embedding dimension: element by element, zeros beyond D
            v.x = src[0];
            if (k + 1 < D) v.y = src[1];
            if (k + 2 < D) v.z = src[2];
            if (k + 3 < D) v.w = src[3];
        }
    }
    out[o] = v;
}

// S[qi][n] = q[qi] . db[n]          (metric IP)
//          = max(0, |q|^2 + |db|^2 - 2 q.db)         (metric L2: squared; the ranking kernels take the root of what they return)
// i0 / j0 are wave-uniform (SGPRs): every store is a scalar row base + ONE per-lane offset (4h rows down, r columns in)
template <bool L2>
__device__ __forceinline__ void scores_store(const f32x16 (&acc)[2][2], float *__restrict__ S, const float *__restrict__ qn,
                                             const float *__restrict__ dbn, int64_t i0, int64_t j0, int Q, int64_t N, int r,
                                             int h, bool interior)
{
    float dn[2] = {0.f, 0.f};
    if (L2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) dn[b] = dbn[min(j0 + b * 32 + r, N - 1)];
    }
    const uint32_t voff = (uint32_t)(4 * h) * (uint32_t)N + (uint32_t)r;   // N <= 2^26: the byte offset fits 32 bits
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        // the norms are fetched with clamped indices, all sixteen loads in flight together (guarded loads would each wait
        // for the one before); only edge tiles guard their stores
        float qv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = i0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            qv[e] = L2 ? qn[min(row, (int64_t)Q - 1)] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool col_ok = interior || j0 + b * 32 + r < N;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row_u = i0 + a * 32 + (e & 3) + 8 * (e >> 2);          // uniform part of the row
                float v = acc[a][b][e];
                if (L2) v = fmaxf(0.f, fmaf(-2.f, v, qv[e] + dn[b]));   // squared: the root is taken of the k results only
                float *base = S + row_u * N + (j0 + b * 32);
                if (col_ok && (interior || row_u + 4 * h < Q)) base[voff] = v;
            }
        }
    }
}

struct ScoreFrag {
    float4 a[2], b[2];   // the 32-row blocks of the wave's 64 x 64 tile
};

__device__ __forceinline__ void scores_load(ScoreFrag &f, const float4 *const (&af)[2], const float4 *const (&bf)[2], int c)
{
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        f.a[blk] = af[blk][(int64_t)c * 64];
        f.b[blk] = bf[blk][(int64_t)c * 64];
    }
}

__device__ __forceinline__ void scores_mma(f32x16 (&acc)[2][2], const ScoreFrag &f)
{
    // consecutive MFMAs go to different accumulators: a dependent pair is four instructions apart
#define WV_SC_STEP(E)                                                                                   \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)         \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[a].E, f.b[b].E, acc[a][b], 0, 0, 0);
    WV_SC_STEP(x) WV_SC_STEP(y) WV_SC_STEP(z) WV_SC_STEP(w)
#undef WV_SC_STEP
}

__global__ __launch_bounds__(256, 3) void k_scores(const float4 *__restrict__ qf, const float4 *__restrict__ dbf,
                                                const float *__restrict__ qn, const float *__restrict__ dbn,
                                                float *__restrict__ S, int Q, int64_t N, int nchunk, int metric)
{
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(wave_id());
    const int r = lane & 31, h = lane >> 5;
    // 128x128 per workgroup, wave (wv>>1, wv&1) owns a 64x64 quadrant
    const int64_t i0 = (int64_t)blockIdx.y * 128 + (wv >> 1) * 64;
    const int64_t j0 = (int64_t)blockIdx.x * 128 + (wv & 1) * 64;
    if (i0 >= Q || j0 >= N) return;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // both images are padded to whole 128-row tiles, so every block a wave touches exists
    const float4 *af[2], *bf[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        af[blk] = qf + ((i0 >> 5) + blk) * nchunk * 64 + lane;
        bf[blk] = dbf + ((j0 >> 5) + blk) * nchunk * 64 + lane;
    }
    // the fragments of chunk c + 1 are in flight while the 16 MFMAs of chunk c issue (pinned with sched_barrier: left
    // alone, the scheduler sinks every load to its first use)
    ScoreFrag f0, f1;
    scores_load(f0, af, bf, 0);
    int c = 0;
    for (; c + 2 < nchunk; c += 2) {
        scores_load(f1, af, bf, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        scores_mma(acc, f0);
        __builtin_amdgcn_sched_barrier(0);
        scores_load(f0, af, bf, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        scores_mma(acc, f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 1 < nchunk) {
        scores_load(f1, af, bf, c + 1);
        scores_mma(acc, f0);
        scores_mma(acc, f1);
    } else {
        scores_mma(acc, f0);
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool interior = i0 + 64 <= Q && j0 + 64 <= N;
    if (metric != WV_METRIC_IP) scores_store<true>(acc, S, qn, dbn, i0, j0, Q, N, r, h, interior);
    else scores_store<false>(acc, S, qn, dbn, i0, j0, Q, N, r, h, interior);
}

// The same tile with the operands shared through LDS: a 128 x 128 workgroup tile needs 4 + 4 row blocks per chunk, but the
// four waves of k_scores fetch 2 + 2 each -- every fragment twice, 0.0625 B/flop from L2.  Here wave w fetches block w of
// A and of B for a stage of four chunks (8 coalesced 1 KB loads, kept in registers while the previous stage computes),
// stores them to one half of a double-buffered LDS ring, and every wave reads its 2 + 2 blocks back as ds_read_b128
// (lane-contiguous: conflict-free).  One barrier per stage of 64 MFMAs.
#ifndef WV_SC_STAGE
#define WV_SC_STAGE 4
#endif
#ifndef WV_SC_WAVES
#define WV_SC_WAVES 2
#endif
constexpr int kScStage = WV_SC_STAGE;                          // chunks of 8 k-values per stage
constexpr int kScStageF4 = 2 * 4 * kScStage * 64;              // float4 per stage: (A, B) x 4 blocks x chunks x lanes

__global__ __launch_bounds__(256, WV_SC_WAVES) void k_scores_lds(const float4 *__restrict__ qf, const float4 *__restrict__ dbf,
                                                       const float *__restrict__ qn, const float *__restrict__ dbn,
                                                       float *__restrict__ S, int Q, int64_t N, int nchunk, int metric)
{
    __shared__ float4 ring[2 * kScStageF4];                    // 64 KB
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(wave_id());
    const int r = lane & 31, h = lane >> 5;
    const int64_t ti = (int64_t)blockIdx.y * 128, tj = (int64_t)blockIdx.x * 128;
    const int wa = wv >> 1, wb = wv & 1;
    const int64_t i0 = ti + wa * 64, j0 = tj + wb * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // this wave's share of a stage: block wv of the A tile and of the B tile (the images are padded to whole tiles and
    // to whole stages, so every fragment exists)
    const float4 *ga = qf + ((ti >> 5) + wv) * nchunk * 64 + lane;
    const float4 *gb = dbf + ((tj >> 5) + wv) * nchunk * 64 + lane;
    float4 *la = ring + (wv * kScStage) * 64 + lane;                       // A[blk][cc][lane]
    float4 *lb = ring + (4 * kScStage + wv * kScStage) * 64 + lane;        // B[blk][cc][lane]
    const float4 *ra = ring + (2 * wa * kScStage) * 64 + lane;
    const float4 *rb = ring + (4 * kScStage + 2 * wb * kScStage) * 64 + lane;
    // (named registers, not arrays: carried around the loop, arrays of float4 were promoted to LDS / scratch by the compiler)
    static_assert(kScStage == 4 || kScStage == 2, "the staging registers below are written out for two or four chunks");
    float4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
#if WV_SC_STAGE == 4
#define WV_SC_FETCH(OFF)                                                                                        \
    pa0 = ga[(OFF)]; pb0 = gb[(OFF)]; pa1 = ga[(OFF) + 64]; pb1 = gb[(OFF) + 64];                               \
    pa2 = ga[(OFF) + 128]; pb2 = gb[(OFF) + 128]; pa3 = ga[(OFF) + 192]; pb3 = gb[(OFF) + 192];
#define WV_SC_STASH(DA, DB)                                                                                     \
    (DA)[0] = pa0; (DB)[0] = pb0; (DA)[64] = pa1; (DB)[64] = pb1;                                               \
    (DA)[128] = pa2; (DB)[128] = pb2; (DA)[192] = pa3; (DB)[192] = pb3;
#else
#define WV_SC_FETCH(OFF) pa0 = ga[(OFF)]; pb0 = gb[(OFF)]; pa1 = ga[(OFF) + 64]; pb1 = gb[(OFF) + 64];
#define WV_SC_STASH(DA, DB) (DA)[0] = pa0; (DB)[0] = pb0; (DA)[64] = pa1; (DB)[64] = pb1;
    (void)pa2; (void)pa3; (void)pb2; (void)pb3;
#endif
    const int nstage = nchunk / kScStage;
    WV_SC_FETCH((int64_t)0)
    WV_SC_STASH(la, lb)
    {
        const int64_t one = (int64_t)min(1, nstage - 1) * kScStage * 64;
        WV_SC_FETCH(one)
    }
    __syncthreads();
    // Staging pipeline (one register set): stage st + 1 travels global -> registers during stage st - 1's MFMAs, is
    // written to the free half of the ring at the START of stage st (that half was last read before the barrier that
    // ended stage st - 1), and the registers are re-issued for stage st + 2 at once: every load has a whole stage of
    // MFMAs to land and cannot be sunk to its use, which is an iteration away.  Refills are unconditional (the last
    // stages fetch the final stage again and store it where nobody reads: a conditional refill costs a vmcnt(0) per load).
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        {
            float4 *da = la + (buf ^ 1) * kScStageF4, *db_ = lb + (buf ^ 1) * kScStageF4;
            WV_SC_STASH(da, db_)
            const int64_t nxt = (int64_t)min(st + 2, nstage - 1) * kScStage * 64;
            WV_SC_FETCH(nxt)
        }
        __builtin_amdgcn_sched_barrier(0);
        const float4 *sa = ra + buf * kScStageF4, *sb = rb + buf * kScStageF4;
        ScoreFrag f0, f1;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f0.a[blk] = sa[(blk * kScStage + 0) * 64];
            f0.b[blk] = sb[(blk * kScStage + 0) * 64];
        }
#pragma unroll
        for (int cc = 0; cc < kScStage; cc += 2) {
            // the next chunk's fragments are read before this chunk's MFMAs issue (pinned: the scheduler sinks them otherwise)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                f1.a[blk] = sa[(blk * kScStage + cc + 1) * 64];
                f1.b[blk] = sb[(blk * kScStage + cc + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            scores_mma(acc, f0);
            __builtin_amdgcn_sched_barrier(0);
            if (cc + 2 < kScStage) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    f0.a[blk] = sa[(blk * kScStage + cc + 2) * 64];
                    f0.b[blk] = sb[(blk * kScStage + cc + 2) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            scores_mma(acc, f1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#undef WV_SC_FETCH
#undef WV_SC_STASH
    if (i0 >= Q || j0 >= N) return;
    const bool interior = i0 + 64 <= Q && j0 + 64 <= N;
    if (metric != WV_METRIC_IP) scores_store<true>(acc, S, qn, dbn, i0, j0, Q, N, r, h, interior);
    else scores_store<false>(acc, S, qn, dbn, i0, j0, Q, N, r, h, interior);
}

__device__ __forceinline__ uint32_t float_to_key(float v, bool descending)
{
    v += 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending float order -> ascending unsigned
    return descending ? ~u : u;
}
__device__ __forceinline__ float key_to_float(uint32_t u, bool descending)
{
    if (descending) u = ~u;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

// One stable counting-sort pass over digit (key >> shift) & 63 of every row.
// first: source is the score matrix (key derived, idx = column); last: only ranks < k are written,
// to idx_out / val_out.
// Intermediate (key, idx) arrays are kept in the column image pos -> [pos % C][pos / C] ("transposed"), so
// that thread t's contiguous range [t*C, (t+1)*C) of the NEXT pass is read with coalesced loads.
// (a device function: k_row_radix below runs the selection and all six passes of a row in ONE launch -- a row never
// leaves its workgroup, so the passes need a barrier between them, not a kernel boundary)
__device__ __forceinline__ void radix_pass(uint32_t *hist, uint32_t *tot, uint32_t *base, const float *__restrict__ S,
                                           const uint2 *src, uint2 *dst, int64_t N, int C, uint32_t c_magic, int shift,
                                           int first, int last, int k, int descending, int sqrt_out,
                                           int32_t *__restrict__ idx_out, float *__restrict__ val_out)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t row = blockIdx.x;
    const int64_t pitch = (int64_t)C * kRadixThreads;   // padded row length of the transposed images
    const float *Srow = S + row * N;
    const uint2 *srow = src + row * pitch;
    uint2 *drow = dst + row * pitch;

    {
        uint4 *h4 = reinterpret_cast<uint4 *>(hist);
        for (int i = tid; i < kRadixBins * kRadixThreads / 4; i += kRadixThreads) h4[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int64_t first_item = (int64_t)tid * C;
    const int nvalid = (int)min((int64_t)C, max((int64_t)0, N - first_item));
    constexpr int UNR = 8;
    auto fetch = [&](int r) -> uint2 {
        if (first) {
            const int64_t item = first_item + r;
            return make_uint2(float_to_key(Srow[min(item, N - 1)], descending), (uint32_t)item);
        }
        return srow[(int64_t)r * kRadixThreads + tid];
    };
    // counting: a column belongs to one thread -> the LDS adds never contend and nothing waits on them
    int r = 0;
    for (; r + UNR <= C; r += UNR) {
        uint2 kv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) kv[u] = fetch(r + u);
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (r + u < nvalid)
                __hip_atomic_fetch_add(&hist[((kv[u].x >> shift) & (kRadixBins - 1)) * kRadixThreads + tid], 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (; r < nvalid; ++r)
        __hip_atomic_fetch_add(&hist[((fetch(r).x >> shift) & (kRadixBins - 1)) * kRadixThreads + tid], 1u,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    for (int b = wv; b < kRadixBins; b += kRadixThreads / 64) {
        const uint4 v = *reinterpret_cast<const uint4 *>(hist + b * kRadixThreads + 4 * lane);
        const uint32_t s = wave_sum_u32(v.x + v.y + v.z + v.w);
        if (lane == 0) tot[b] = s;
    }
    __syncthreads();
    if (wv == 0) {
        const uint32_t t = tot[lane];
        base[lane] = wave_incl_scan_u32(t) - t;
    }
    __syncthreads();
    for (int b = wv; b < kRadixBins; b += kRadixThreads / 64) {
        uint4 v = *reinterpret_cast<const uint4 *>(hist + b * kRadixThreads + 4 * lane);
        const uint32_t s = v.x + v.y + v.z + v.w;
        const uint32_t excl = wave_incl_scan_u32(s) - s + base[b];
        uint4 o;
        o.x = excl; o.y = excl + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
        *reinterpret_cast<uint4 *>(hist + b * kRadixThreads + 4 * lane) = o;
    }
    __syncthreads();
    auto place = [&](const uint2 &kv, uint32_t pos) {
        if (last) {
            if (pos < (uint32_t)k) {
                idx_out[row * k + pos] = (int32_t)kv.y;
                const float v = key_to_float(kv.x, descending);
                val_out[row * k + pos] = sqrt_out ? sqrtf(v) : v;
            }
        } else {
            const uint32_t tq = C == 1 ? pos : __umulhi(pos, c_magic);   // pos / C
            drow[(int64_t)(pos - tq * C) * kRadixThreads + tq] = kv;
        }
    };
    for (r = 0; r + UNR <= C; r += UNR) {
        uint2 kv[UNR];
        uint32_t pos[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) kv[u] = fetch(r + u);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            pos[u] = 0xffffffffu;
            if (r + u < nvalid)
                pos[u] = __hip_atomic_fetch_add(&hist[((kv[u].x >> shift) & (kRadixBins - 1)) * kRadixThreads + tid], 1u,
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (pos[u] != 0xffffffffu) place(kv[u], pos[u]);
    }
    for (; r < nvalid; ++r) {
        const uint2 kv = fetch(r);
        place(kv, __hip_atomic_fetch_add(&hist[((kv.x >> shift) & (kRadixBins - 1)) * kRadixThreads + tid], 1u,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    __syncthreads();   // the next pass zeroes the columns and reads what this one stored (same workgroup, same CU)
}

// Selection before sorting (k <= N/2): the full LSD sort moves every one of the N (key, index) pairs six times,
// but only the k best are wanted.  One workgroup per row finds the k-th smallest key tau by radix selection on the
// order-preserving 32-bit key (three histogram passes over 11 + 11 + 10 bits, coalesced reads of the score row),
// then compacts -- in index order, i.e. stably -- the items with key < tau plus the first few items with key == tau
// that fill up to exactly k, straight into the column image the radix passes read.  The six passes then sort k
// items instead of N.  Same result as the full sort: ascending (key, index).
constexpr int kSelBins = 2048;
__device__ __forceinline__ void select_compact(uint32_t *hist, uint32_t *wpart, uint32_t *sel,
                                               const float *__restrict__ S, uint2 *dst, int64_t N, int k, int Ck,
                                               uint32_t ck_magic, int descending)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t row = blockIdx.x;
    const float *Srow = S + row * N;
    uint2 *drow = dst + row * (int64_t)Ck * kRadixThreads;
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)k;   // looking for the remaining-th smallest among the matches
    const int shifts[3] = {21, 10, 0};
    const int nbins[3] = {2048, 2048, 1024};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = nbins[pass];
        for (int i = tid; i < kSelBins; i += 256) hist[i] = 0;
        __syncthreads();
        for (int64_t i = tid; i < N; i += 256) {
            const uint32_t key = float_to_key(Srow[i], descending);
            if ((key & mask) == prefix)
                __hip_atomic_fetch_add(&hist[(key >> shift) & (uint32_t)(nb - 1)], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        // bin where the running count reaches `remaining`: thread t owns bins [t*per, (t+1)*per)
        const int per = nb / 256;
        uint32_t mine = 0;
        for (int j = 0; j < per; ++j) mine += hist[tid * per + j];
        const uint32_t incl = wave_incl_scan_u32(mine);
        if (lane == 63) wpart[wv] = incl;
        __syncthreads();
        uint32_t before = incl - mine;
        for (int w2 = 0; w2 < wv; ++w2) before += wpart[w2];
        if (before < remaining && remaining <= before + mine) {   // exactly one thread
            uint32_t run = before;
            for (int j = 0; j < per; ++j) {
                const uint32_t c = hist[tid * per + j];
                if (remaining <= run + c) { sel[0] = (uint32_t)(tid * per + j); sel[1] = run; break; }
                run += c;
            }
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        mask |= (uint32_t)(nb - 1) << shift;
        remaining -= sel[1];
        __syncthreads();
    }
    const uint32_t tau = prefix, need_eq = remaining;             // take every key < tau and the first need_eq keys == tau
    // stable compaction, 256 consecutive items per step
    uint32_t base_less = 0, base_eq = 0;
    for (int64_t i0 = 0; i0 < N; i0 += 256) {
        const int64_t i = i0 + tid;
        uint32_t key = 0xffffffffu;
        bool less = false, eq = false;
        if (i < N) {
            key = float_to_key(Srow[i], descending);
            less = key < tau;
            eq = key == tau;
        }
        const uint64_t ml = __ballot(less), me = __ballot(eq);
        if (lane == 0) { wpart[wv] = (uint32_t)__popcll(ml); wpart[4 + wv] = (uint32_t)__popcll(me); }
        __syncthreads();
        uint32_t lb = base_less + (uint32_t)mbcnt(ml), eb = base_eq + (uint32_t)mbcnt(me);
        for (int w2 = 0; w2 < wv; ++w2) { lb += wpart[w2]; eb += wpart[4 + w2]; }
        const uint32_t tot_l = wpart[0] + wpart[1] + wpart[2] + wpart[3], tot_e = wpart[4] + wpart[5] + wpart[6] + wpart[7];
        if (less || (eq && eb < need_eq)) {
            const uint32_t pos = lb + min(eb, need_eq);           // survivors before this one, in index order
            const uint32_t tq = Ck == 1 ? pos : __umulhi(pos, ck_magic);   // pos / Ck
            drow[(int64_t)(pos - tq * Ck) * kRadixThreads + tq] = make_uint2(key, (uint32_t)i);
        }
        base_less += tot_l;
        base_eq += tot_e;
        __syncthreads();
    }
}

// The radix ranking of one row, one launch: [selection of the k best] + six stable counting-sort passes.
// todo != null: only rows k_row_topk (below) handed over; every other workgroup returns at once.
__global__ __launch_bounds__(kRadixThreads) void k_row_radix(const float *__restrict__ S, uint2 *bufA, uint2 *bufB,
                                                             int64_t N, int C, uint32_t c_magic, int k, int Ck,
                                                             uint32_t ck_magic, int select_first, int descending,
                                                             int sqrt_out, int32_t *__restrict__ idx_out,
                                                             float *__restrict__ val_out,
                                                             const uint8_t *__restrict__ todo)
{
    if (todo && !todo[blockIdx.x]) return;
    __shared__ uint32_t hist[kRadixBins * kRadixThreads];  // 64 KB (the selection's 2048 bins live in its head)
    __shared__ uint32_t tot[kRadixBins];
    __shared__ uint32_t base[kRadixBins];
    constexpr int npass = 32 / kRadixBits + (32 % kRadixBits != 0);
    if (select_first) {
        // radix-select the k best of the row, then sort only those (the images keep the N-sized pitch)
        select_compact(hist, tot, base, S, bufB, N, k, Ck, ck_magic, descending);
        __syncthreads();
        for (int p = 0; p < npass; ++p)
            radix_pass(hist, tot, base, S, (p & 1) ? bufA : bufB, (p & 1) ? bufB : bufA, (int64_t)k, Ck, ck_magic,
                       p * kRadixBits, 0, p == npass - 1, k, descending, sqrt_out, idx_out, val_out);
        return;
    }
    for (int p = 0; p < npass; ++p)
        radix_pass(hist, tot, base, S, (p & 1) ? bufA : bufB, (p & 1) ? bufB : bufA, N, C, c_magic, p * kRadixBits,
                   p == 0, p == npass - 1, k, descending, sqrt_out, idx_out, val_out);
}

// ---- one kernel per row for k <= 15,360: the list is selected, sorted and written without leaving the CU ----------
// k_select_compact + six k_radix_pass launches move every survivor through global memory seven times with scattered
// 8-byte stores (0.87 ms of the 1.16 ms of a 2048 x 25,000, k = 5000 search), and the selection's 11-bit histograms
// of raw key bits pile a whole row onto the few bins of its sign / exponent.  Here the digit is a VALUE bin instead:
//   bin(w) = clamp(floor(64 + (w - lo) * 3968 / (hi - lo)), 0, 4095),   w = the score in ascending rank order
//   (v for L2, -v for IP), [lo, hi] = min / max of a strided 4096-item sample of the row.
// Every step of bin() is monotone non-decreasing under IEEE rounding, so bin(x) < bin(y) implies x < y and equal
// keys share a bin -- for ANY lo / hi (a poor range costs time, never correctness).  One histogram pass finds the
// bin b* that holds the k-th item; a second pass drops every item of bins <= b* into its bin's slot range of an LDS
// list (order inside a bin arbitrary); each item then counts the items of its own bin that precede it as
// (key, index) pairs -- a handful for smooth scores -- which is its final rank; the list is permuted in place and
// its first k entries leave as two coalesced streams.  Rows this cannot take (a bin holding more than 512
// survivors, NaNs, a boundary bin that overflows the list: ties en masse, constant rows) set todo[row] and the radix
// kernels above rank exactly those rows; they return at once for all others.
constexpr int kTkThreads = 1024;
constexpr int kTkBins = 4096;
constexpr int kTkSlack = 1024;        // list entries beyond k for the rest of the boundary bin
constexpr int kTkCapMax = 16384;      // 16 entries per thread are held in registers across the in-place permutation
constexpr int kTkGroupMax = 512;
constexpr int kTkSample = 4096;
#ifndef WV_TK_UNROLL
#define WV_TK_UNROLL 8
#endif
constexpr int kTkUnroll = WV_TK_UNROLL;

__device__ __forceinline__ float tk_order_value(float v, bool descending)
{
    v += 0.0f;   // -0 -> +0, as float_to_key
    return descending ? -v : v;
}
__device__ __forceinline__ int tk_bin(float w, float lo, float scale)
{
    return (int)fminf(fmaxf(fmaf(w - lo, scale, 64.f), 0.f), (float)(kTkBins - 1));
}

static size_t tk_lds_bytes(int cap) { return (size_t)(kTkBins + 32 + 32 + 8) * 4 + (size_t)cap * 8; }

// exclusive prefix over the bins (thread t owns bins 4t .. 4t+3): returns the prefix of bin 4t, the four counts in c
__device__ __forceinline__ uint32_t tk_scan_bins(const uint32_t *cnt, uint32_t *wpart, uint4 &c)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    c = *reinterpret_cast<const uint4 *>(cnt + 4 * tid);
    const uint32_t sum = c.x + c.y + c.z + c.w;
    const uint32_t incl = wave_incl_scan_u32(sum);
    if (lane == 63) wpart[wv] = incl;
    __syncthreads();
    uint32_t e0 = incl - sum;
    for (int w2 = 0; w2 < wv; ++w2) e0 += wpart[w2];
    return e0;
}

enum { TK_BSTAR = 0, TK_BEFORE = 1, TK_COUNT = 2, TK_NAN_SEEN = 3, TK_GMAX = 4, TK_BLO = 5, TK_FOUND = 6 };   // sel[]

// Value-bin histogram of the whole row (cursor[] zeroed by the caller), exclusive prefix in place, the bin b* where the
// running count reaches k -> true when the list fits (b* found, at most cap items up to b*, no bin up to b* beyond
// kTkGroupMax).  If the bins of the sampled range are too coarse where it matters -- outliers stretched [lo, hi] -- the
// histogram still says where the first k items lie, and the caller makes ONE second attempt that spreads [the bin holding
// item k / 16, b*] of the first over all 4096 bins (SECOND: items far above pile up in the last bin and are not counted; if
// item k lands there after all, the row goes to the radix kernel).  Any range is a correct binning.
template <bool SECOND>
__device__ __forceinline__ bool tk_histogram(const float *__restrict__ Srow, int64_t N, int k, int cap, bool desc, float lo,
                                             float scale, uint32_t *cursor, uint32_t *wpart, uint32_t *sel)
{
    constexpr int UNR = kTkUnroll;
    const int tid = threadIdx.x, lane = lane_id();
    bool nan_seen = false;
    for (int64_t i0 = tid; i0 < N; i0 += (int64_t)kTkThreads * UNR) {
        float v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * kTkThreads;
            v[u] = Srow[min(i, N - 1)];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (i0 + (int64_t)u * kTkThreads >= N) break;
            const float w = tk_order_value(v[u], desc);
            nan_seen |= w != w;
            const int b = tk_bin(w, lo, scale);
            if (!SECOND || b < kTkBins - 1)
                __hip_atomic_fetch_add(&cursor[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (nan_seen) sel[TK_NAN_SEEN] = 1;
    __syncthreads();
    uint4 c;
    const uint32_t e0 = tk_scan_bins(cursor, wpart, c), e1 = e0 + c.x, e2 = e1 + c.y, e3 = e2 + c.z;
    *reinterpret_cast<uint4 *>(cursor + 4 * tid) = make_uint4(e0, e1, e2, e3);
    const uint32_t kk = (uint32_t)k, klo = max(1u, min(kk / 16u, 256u));
    const uint32_t e[4] = {e0, e1, e2, e3}, cc[4] = {c.x, c.y, c.z, c.w};
    uint32_t g = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (e[b] < kk && kk <= e[b] + cc[b]) {
            sel[TK_BSTAR] = 4 * tid + b; sel[TK_BEFORE] = e[b]; sel[TK_COUNT] = cc[b]; sel[TK_FOUND] = 1;
        }
        if (!SECOND && e[b] < klo && klo <= e[b] + cc[b]) sel[TK_BLO] = 4 * tid + b;
        g = max(g, e[b] < kk ? cc[b] : 0u);
    }
    const uint32_t gw = (uint32_t)__reduce_max_sync(~0ull, g);
    if (lane == 0) __hip_atomic_fetch_max(&sel[TK_GMAX], gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    return sel[TK_FOUND] && sel[TK_BEFORE] + sel[TK_COUNT] <= (uint32_t)cap && sel[TK_GMAX] <= (uint32_t)kTkGroupMax;
}

// Occupancy: two workgroups of 1024 threads per CU = 8 waves per SIMD need <= 64 VGPRs AND <= 80 SGPRs (the 800-entry scalar
// file holds 16 more per wave than the kernel asks for: at 96 a SIMD takes 7 waves, i.e. ONE workgroup per CU, and every phase
// of this latency-bound kernel ran 1.5 x longer -- measured when the second binning attempt pushed the count from 80 to 96).
// The bound makes the compiler keep to 78 (two scalars spilled to a VGPR's lanes).  JMAX = 16 needs 147 KB of LDS: one per CU anyway.
template <int JMAX>
__global__ __launch_bounds__(kTkThreads, (JMAX <= 8 ? 8 : 4)) void k_row_topk(const float *__restrict__ S, int64_t N, int k, int cap,
                                                         int descending, int32_t *__restrict__ idx_out,
                                                         float *__restrict__ val_out, uint8_t *__restrict__ todo,
                                                         int force_radix, int sqrt_out)
{
    extern __shared__ uint4 tk_sm4[];
    uint32_t *cursor = reinterpret_cast<uint32_t *>(tk_sm4);   // [4096] counts -> exclusive prefix -> running cursor
    uint32_t *wpart = cursor + kTkBins;                         // [32]   wave totals of the scan
    float *fpart = reinterpret_cast<float *>(wpart + 32);       // [32]   wave minima, wave maxima
    uint32_t *sel = reinterpret_cast<uint32_t *>(fpart + 32);   // [8]    see the enum
    uint64_t *A = reinterpret_cast<uint64_t *>(sel + 8);        // [cap]  (key << 32 | index)
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t row = blockIdx.x;
    const float *Srow = S + row * N;
    const bool desc = descending != 0;
#ifdef WV_TK_STOPS   // phase timing by truncation (tools/knn_phases.py): the kernel ends after phase force_radix >> 8
    const int stop_at = force_radix >> 8;
    force_radix &= 0xff;
#define TK_STOP(n) if (stop_at == (n)) return
#else
#define TK_STOP(n)
#endif

    // range of the row from a sample: 64 runs of 64 consecutive scores spread over the row (the whole row if short)
    float mn = __builtin_inff(), mx = -__builtin_inff();
    {
        const bool sampled = N > kTkSample;
        const int64_t stride = N / 64;
#pragma unroll
        for (int j = 0; j < kTkSample / kTkThreads; ++j) {
            const int s = tid + kTkThreads * j;
            const int64_t i = sampled ? (int64_t)(s >> 6) * stride + (s & 63) : (int64_t)s;
            if (i < N) {
                const float w = tk_order_value(Srow[i], desc);
                mn = fminf(mn, w);
                mx = fmaxf(mx, w);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, d, 64));
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    if (lane == 0) { fpart[wv] = mn; fpart[16 + wv] = mx; }
    *reinterpret_cast<uint4 *>(cursor + 4 * tid) = make_uint4(0, 0, 0, 0);
    if (tid < 8) sel[tid] = 0;
    __syncthreads();
    float lo = fpart[0], hi = fpart[16];
#pragma unroll
    for (int w2 = 1; w2 < kTkThreads / 64; ++w2) {
        lo = fminf(lo, fpart[w2]);
        hi = fmaxf(hi, fpart[16 + w2]);
    }
    float scale = (float)(kTkBins - 128) / (hi - lo);
    if (!(scale > 0.f)) scale = 0.f;                 // constant sample, infinite range, NaN: everything in one bin
    TK_STOP(1);

    // pass 1: value-bin histogram and the bin b* of the k-th item; one second attempt with a range taken from the first
    // histogram when outliers made the sampled range too coarse (tk_histogram)
    constexpr int UNR = kTkUnroll;
    bool fits = tk_histogram<false>(Srow, N, k, cap, desc, lo, scale, cursor, wpart, sel);
    TK_STOP(2);
    if (sel[TK_NAN_SEEN] || force_radix || (!fits && !(scale > 0.f))) {
        if (tid == 0) todo[row] = 1;
        return;
    }
    if (!fits) {
        // second attempt: [lower edge of bin b_lo, upper edge of bin b*] in the order value's space
        const float inv = 1.f / scale;
        const float nlo = fmaf((float)((int)sel[TK_BLO] - 64), inv, lo), nhi = fmaf((float)((int)sel[TK_BSTAR] + 1 - 64), inv, lo);
        __syncthreads();                                  // everybody has read sel[]
        lo = nlo;
        scale = (float)(kTkBins - 128) / (nhi - nlo);
        if (!(scale > 0.f) || !(scale < __builtin_inff())) {
            if (tid == 0) todo[row] = 1;
            return;
        }
        *reinterpret_cast<uint4 *>(cursor + 4 * tid) = make_uint4(0, 0, 0, 0);
        if (tid < 8) sel[tid] = 0;
        __syncthreads();
        if (!tk_histogram<true>(Srow, N, k, cap, desc, lo, scale, cursor, wpart, sel)) {
            if (tid == 0) todo[row] = 1;
            return;
        }
    }
    const int bstar = (int)sel[TK_BSTAR];
    const uint32_t n_tot = sel[TK_BEFORE] + sel[TK_COUNT];
    if (tid == 0) todo[row] = 0;
    TK_STOP(3);

    // pass 2: every item of a bin <= b* takes the next slot of its bin; afterwards cursor[b] is the END of bin b, i.e. the
    // start of bin b + 1
    for (int64_t i0 = tid; i0 < N; i0 += (int64_t)kTkThreads * UNR) {
        float v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * kTkThreads;
            v[u] = Srow[min(i, N - 1)];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + (int64_t)u * kTkThreads;
            if (i >= N) break;
            const int b = tk_bin(tk_order_value(v[u], desc), lo, scale);
            if (b <= bstar)
                A[__hip_atomic_fetch_add(&cursor[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)] =
                    ((uint64_t)float_to_key(v[u], desc) << 32) | (uint32_t)i;
        }
    }
    __syncthreads();
    TK_STOP(4);

    // rank inside the bin = number of (key, index) pairs of the same bin that come first (four per trip: the reads of a
    // trip are independent, a lane's neighbours are in the same bin and read the same addresses)
    uint64_t item[JMAX];
    uint32_t pos[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const uint32_t i = (uint32_t)tid + (uint32_t)(kTkThreads * j);
        pos[j] = 0xffffffffu;
        if (i < n_tot) {
            const uint64_t e = A[i];
            const int b = tk_bin(tk_order_value(key_to_float((uint32_t)(e >> 32), desc), desc), lo, scale);
            const uint32_t s = b ? cursor[b - 1] : 0u, end = cursor[b];
            uint32_t before = s;
            for (uint32_t p = s; p < end; p += 4) {
                const uint32_t last = end - 1;
                const uint64_t a0 = A[p], a1 = A[min(p + 1, last)], a2 = A[min(p + 2, last)], a3 = A[min(p + 3, last)];
                before += (a0 < e) + (p + 1 < end && a1 < e) + (p + 2 < end && a2 < e) + (p + 3 < end && a3 < e);
            }
            item[j] = e;
            pos[j] = before;
        }
    }
    __syncthreads();
    TK_STOP(5);
#pragma unroll
    for (int j = 0; j < JMAX; ++j)
        if (pos[j] != 0xffffffffu) A[pos[j]] = item[j];
    __syncthreads();
    for (int i = tid; i < k; i += kTkThreads) {
        const uint64_t e = A[i];
        idx_out[row * k + i] = (int32_t)(uint32_t)e;
        const float v = key_to_float((uint32_t)(e >> 32), desc);
        val_out[row * k + i] = sqrt_out ? sqrtf(v) : v;
    }
}

template <int JMAX>
static int launch_row_topk(const float *S, int64_t N, int k, int cap, int descending, int32_t *idx, float *val,
                           uint8_t *todo, int force_radix, int sqrt_out, int rows, hipStream_t st)
{
    const size_t lds = tk_lds_bytes(cap);
    auto kern = k_row_topk<JMAX>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "knn_float: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(kTkThreads), lds, st, S, N, k, cap, descending, idx, val, todo,
                       force_radix, sqrt_out);
    return WV_OK;
}

// The ranking stage: the k best of every row of S [qc][N] in ascending (key, column) order.  Rows the one-kernel ranking takes
// never reach the radix kernel (its workgroups return at once on todo[row] == 0).
static int rank_rows(const float *S, int qc, int64_t N, int k, int descending, int sqrt_out, int32_t *idx, float *val,
                     uint8_t *todo_buf, uint2 *bufA, uint2 *bufB, hipStream_t st)
{
    const int C = (int)ceil_div(N, kRadixThreads);
    const uint32_t c_magic = C <= 1 ? 0u : (uint32_t)(((1ull << 32) + C - 1) / C);   // exact for pos < 2^32 / C
    const uint8_t *todo = nullptr;
    if (k + kTkSlack <= kTkCapMax && !::wv::tune("WV_KNN_RADIX_ONLY")) {
        const int cap = (int)std::min<int64_t>(kTkCapMax, align_up(k + kTkSlack, kTkThreads));
        int force = ::wv::tune("WV_KNN_FORCE_TODO") ? 1 : 0;   // diagnostic build: every row takes both kernels' hand-over
        if (const char *stop = ::wv::tune("WV_TK_STOP")) force |= atoi(stop) << 8;
        const int J = cap / kTkThreads;
        int rc;
        if (J <= 2) rc = launch_row_topk<2>(S, N, k, cap, descending, idx, val, todo_buf, force, sqrt_out, qc, st);
        else if (J <= 4) rc = launch_row_topk<4>(S, N, k, cap, descending, idx, val, todo_buf, force, sqrt_out, qc, st);
        else if (J <= 8) rc = launch_row_topk<8>(S, N, k, cap, descending, idx, val, todo_buf, force, sqrt_out, qc, st);
        else rc = launch_row_topk<16>(S, N, k, cap, descending, idx, val, todo_buf, force, sqrt_out, qc, st);
        if (rc != WV_OK) return rc;
        todo = todo_buf;
    }
    const int select_first = (int64_t)k * 2 <= N && !::wv::tune("WV_KNN_FULLSORT");
    const int Ck = (int)ceil_div(k, kRadixThreads);
    const uint32_t ck_magic = Ck <= 1 ? 0u : (uint32_t)(((1ull << 32) + Ck - 1) / Ck);
    hipLaunchKernelGGL(k_row_radix, dim3(qc), dim3(kRadixThreads), 0, st, S, bufA, bufB, N, C, c_magic, k, Ck, ck_magic,
                       select_first, descending, sqrt_out, idx, val, todo);
    return WV_OK;
}

// rows per launch of wv_rank_scores: the radix images (16 bytes per score) stay within about 1 GiB
static int64_t rank_chunk_rows(int Q, int64_t N)
{
    const int64_t per_row = ceil_div(N, kRadixThreads) * kRadixThreads * 16;
    return std::min<int64_t>(std::max<int64_t>((1ll << 30) / std::max<int64_t>(per_row, 1), 1), Q);
}

static int64_t knn_chunk_rows(int Q, int64_t N)
{
    // bound the scratch to about 2 GiB: per row N * (4 + 8 + 8) bytes
    const int64_t per_row = N * 4 + ceil_div(N, kRadixThreads) * kRadixThreads * 16;
    int64_t rows = (2ll << 30) / std::max<int64_t>(per_row, 1);
    rows = std::max<int64_t>(rows, 1);
    return std::min<int64_t>(rows, Q);
}

}  // namespace wv

using namespace wv;

extern "C" size_t wv_knn_float_workspace_bytes(int Q, int64_t N, int D, int k)
{
    (void)k;
    if (Q <= 0 || N <= 0) return 0;
    const int64_t rows = knn_chunk_rows(Q, N);
    const int64_t pitch = ceil_div(N, kRadixThreads) * kRadixThreads;
    return (size_t)(rows * (N * 4 + pitch * 16) + (align_up(Q, 64) + align_up(N, 64)) * 4 + align_up(rows, 256) +
                    (align_up(rows, 128) + align_up(N, 128)) * align_up(D, 8 * kScStage) * 4 + 1024);
}

extern "C" int wv_knn_float(const float *q, const float *db, int Q, int64_t N, int D, int metric, int k,
                            int32_t *idx, float *val, void *workspace, size_t workspace_bytes,
                            void *stream)
{
    WV_REQUIRE(q && db && idx && val, "knn_float: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1 && D >= 1, "knn_float: bad shape Q=%d N=%lld D=%d", Q, (long long)N, D);
    WV_REQUIRE(metric == WV_METRIC_IP || metric == WV_METRIC_L2 || metric == WV_METRIC_L2_SQUARED, "knn_float: metric %d",
               metric);
    WV_REQUIRE(k >= 1 && k <= N, "knn_float: k=%d must be in [1, N=%lld] (torch.topk raises too)", k,
               (long long)N);
    WV_REQUIRE(N <= (1ll << 26), "knn_float: N=%lld above the supported 2^26 rows", (long long)N);
    const size_t need = wv_knn_float_workspace_bytes(Q, N, D, k);
    if (!workspace || workspace_bytes < need)
        WV_FAIL(WV_ENOMEM, "knn_float: workspace %zu < %zu bytes", workspace_bytes, need);
    if (Q == 0) return WV_OK;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = knn_chunk_rows(Q, N);
    char *w = (char *)workspace;
    const int64_t pitch = ceil_div(N, kRadixThreads) * kRadixThreads;
    float *S = (float *)w;                 w += align_up(rows * N * 4, 256);
    uint2 *bufA = (uint2 *)w;              w += rows * pitch * 8;
    uint2 *bufB = (uint2 *)w;              w += rows * pitch * 8;
    float *qn = (float *)w;                w += align_up(Q, 64) * 4;
    float *dbn = (float *)w;               w += align_up(N, 64) * 4;
    uint8_t *todo_buf = (uint8_t *)w;      w += align_up(rows, 256);   // [rows] 1 = k_row_topk left the row to the radix kernel
    const int nchunk = (int)align_up(ceil_div(D, 8), kScStage);   // chunks of 8 k-values, whole stages (zero-padded)
    // fragment images, padded to whole 128-row tiles: the database once, the queries of a chunk per chunk
    float4 *qf = (float4 *)w;              w += align_up(rows, 128) * nchunk * 32;   // bytes: rows * nchunk * 8 floats
    float4 *dbf = (float4 *)w;
    {
        const int64_t tn = align_up(N, 128) / 32 * nchunk * 64;
        hipLaunchKernelGGL(k_pack_fragments, dim3((unsigned)ceil_div(tn, 256)), dim3(256), 0, st, db, dbf, N, D, nchunk, tn);
    }
    if (metric != WV_METRIC_IP) {
        hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)ceil_div(Q, 4)), dim3(256), 0, st, q, (int64_t)Q, D, qn);
        hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, db, N, D, dbn);
    }
    const int descending = metric == WV_METRIC_IP;
    const int sqrt_out = metric == WV_METRIC_L2;   // L2 rows are ranked on SQUARED distances (what faiss ranks on; the same
                                                   // order up to ties the rounding of the root creates); the k results get the root
    for (int64_t q0 = 0; q0 < Q; q0 += rows) {
        const int qc = (int)std::min<int64_t>(rows, Q - q0);
        dim3 grid((unsigned)ceil_div(N, 128), (unsigned)ceil_div(qc, 128));
        const int64_t tq = align_up(qc, 128) / 32 * nchunk * 64;
        hipLaunchKernelGGL(k_pack_fragments, dim3((unsigned)ceil_div(tq, 256)), dim3(256), 0, st, q + q0 * D, qf, (int64_t)qc, D,
                           nchunk, tq);
        if (::wv::tune("WV_KNN_SCORES_DIRECT"))
            hipLaunchKernelGGL(k_scores, grid, dim3(256), 0, st, qf, dbf, qn + q0, dbn, S, qc, N, nchunk, metric);
        else
            hipLaunchKernelGGL(k_scores_lds, grid, dim3(256), 0, st, qf, dbf, qn + q0, dbn, S, qc, N, nchunk, metric);
        const int rc = rank_rows(S, qc, N, k, descending, sqrt_out, idx + q0 * k, val + q0 * k, todo_buf, bufA, bufB, st);
        if (rc != WV_OK) return rc;
    }
    WV_CHECK_LAUNCH("knn_float");
    return WV_OK;
}

extern "C" size_t wv_rank_scores_workspace_bytes(int Q, int64_t N, int k)
{
    (void)k;
    if (Q <= 0 || N <= 0) return 0;
    const int64_t rows = rank_chunk_rows(Q, N);
    return (size_t)(rows * ceil_div(N, kRadixThreads) * kRadixThreads * 16 + align_up(rows, 256) + 256);
}

extern "C" int wv_rank_scores(const float *S, int Q, int64_t N, int k, int flags, int32_t *idx, float *val, void *workspace,
                              size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(S && idx && val, "rank_scores: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 1, "rank_scores: bad shape Q=%d N=%lld", Q, (long long)N);
    WV_REQUIRE(k >= 1 && k <= N, "rank_scores: k=%d must be in [1, N=%lld]", k, (long long)N);
    WV_REQUIRE(N <= (1ll << 26), "rank_scores: N=%lld above the supported 2^26 columns", (long long)N);
    WV_REQUIRE((flags & ~(WV_RANK_DESCENDING | WV_RANK_SQRT)) == 0, "rank_scores: flags %d", flags);
    const size_t need = wv_rank_scores_workspace_bytes(Q, N, k);
    if (!workspace || workspace_bytes < need)
        WV_FAIL(WV_ENOMEM, "rank_scores: workspace %zu < %zu bytes", workspace_bytes, need);
    if (Q == 0) return WV_OK;
    const int64_t rows = rank_chunk_rows(Q, N), pitch = ceil_div(N, kRadixThreads) * kRadixThreads;
    char *w = (char *)workspace;
    uint2 *bufA = (uint2 *)w;              w += rows * pitch * 8;
    uint2 *bufB = (uint2 *)w;              w += rows * pitch * 8;
    uint8_t *todo_buf = (uint8_t *)w;
    for (int64_t q0 = 0; q0 < Q; q0 += rows) {
        const int qc = (int)std::min<int64_t>(rows, Q - q0);
        const int rc = rank_rows(S + q0 * N, qc, N, k, (flags & WV_RANK_DESCENDING) != 0, (flags & WV_RANK_SQRT) != 0, idx + q0 * k,
                                 val + q0 * k, todo_buf, bufA, bufB, (hipStream_t)stream);
        if (rc != WV_OK) return rc;
    }
    WV_CHECK_LAUNCH("rank_scores");
    return WV_OK;
}

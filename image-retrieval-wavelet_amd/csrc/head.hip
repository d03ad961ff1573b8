// Band-attention pooling head + hashing tail (eval mode) for gfx950.
//
// Reference: CrossAttentionBottleneckHeadAdvanced.forward
// (/root/reference/main/models/multi_dino_attention.py:1111-1141): learned query tokens attend
// over the 4 band tokens through nn.MultiheadAttention, then LN, a GELU MLP, a read-out Linear
// and LN; SharedDinoHashing's tail (:829-833): hash_fc -> BatchNorm1d(eval) -> sign.
//
// Every dense contraction (K/V in_proj, attention out_proj, both MLP layers, the read-out) runs
// on the matrix cores with v_mfma_f32_32x32x2_f32 -- fp32 in, fp32 accumulate, bit-for-bit an
// fmaf chain -- so the result stays within fp32 rounding of the reference's fp32 ATen path
// instead of trading precision for bf16 MFMA rate.  The 4-token softmax / context step is a small
// VALU kernel (QK^T is Nq x 4 per head).  This file holds the general path -- one launch per stage,
// intermediates in a caller workspace -- for any shape, plus the read-out product, LayerNorm 2 and the
// hashing tail every configuration uses.  For the shapes the reference runs (E = 384, 4 band tokens,
// 4 or 8 queries) and batches that fill the chip, everything before the read-out is ONE kernel
// (head_front.hip) fed by a weight stream wv_band_attn_prepare makes once per parameter update; the
// batch-invariant query projection is likewise made once (wv_band_attn_qproj) and cached by the module.
#include "common.hpp"

namespace wv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;   // native vector: arrays of it stay in registers

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_ADD_ROW = 2, EPI_ADD_BCAST = 3 };

// C[M][N] = epi(A[M][K] . W[N][K]^T + bias[N])
//   EPI_ADD_ROW   : + R[m][n]                    (residual of the same shape)
//   EPI_ADD_BCAST : + R[(m % rmod)][n]           (query tokens broadcast over the batch)
// Wave tile = (32*TM) x (32*TN); workgroup = 2 x 2 waves.  Same operand scheme as k_scores
// (knn_float.hip): lane l feeds row l&31, lane half h covers k in [8c+4h, 8c+4h+4).
template <int TM, int TN, int EPI>
__global__ __launch_bounds__(256) void k_gemm_nt(const float *__restrict__ A, const float *__restrict__ W,
                                                 const float *__restrict__ bias,
                                                 const float *__restrict__ R, int rmod,
                                                 float *__restrict__ C, int M, int N, int K)
{
    const int lane = lane_id(), wv = wave_id();
    const int r = lane & 31, h = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.y * (64 * TM) + (wv >> 1) * (32 * TM);
    const int64_t j0 = (int64_t)blockIdx.x * (64 * TN) + (wv & 1) * (32 * TN);
    if (i0 >= M || j0 >= N) return;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    const float *arow[TM], *brow[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) arow[a] = A + min(i0 + a * 32 + r, (int64_t)M - 1) * K;
#pragma unroll
    for (int b = 0; b < TN; ++b) brow[b] = W + min(j0 + b * 32 + r, (int64_t)N - 1) * K;

    for (int k = 0; k < K; k += 8) {  // K % 8 == 0 (checked on the host)
        float4 av[TM], bv[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) av[a] = *reinterpret_cast<const float4 *>(arow[a] + k + 4 * h);
#pragma unroll
        for (int b = 0; b < TN; ++b) bv[b] = *reinterpret_cast<const float4 *>(brow[b] + k + 4 * h);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].z, bv[b].z, acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].w, bv[b].w, acc[a][b], 0, 0, 0);
            }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int64_t col = j0 + b * 32 + r;
            if (col >= N) continue;
            const float bs = bias ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = i0 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = acc[a][b][e] + bs;
                if (EPI == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if (EPI == EPI_ADD_ROW) v += R[row * N + col];
                if (EPI == EPI_ADD_BCAST) v += R[(int64_t)((uint32_t)row % (uint32_t)rmod) * N + col];   // rows < 2^31 (host check)
                C[row * N + col] = v;
            }
        }
}

// LDS-tiled variant for K % 32 == 0: block tile BM x BN, K step 32, two LDS stages (global loads of step
// i+1 are in flight while step i feeds the matrix cores).  4 waves as 2 x 2; wave tile (BM/2) x (BN/2) in
// 32 x 32 MFMA blocks.  Rows are padded to 36 floats: lanes = rows then hit 16 distinct 4-bank groups per
// ds_read_b128 (conflict-free).  Same k <-> (step, lane half) bijection as k_gemm_nt.
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_lds(const float *__restrict__ A, const float *__restrict__ W,
                                                  const float *__restrict__ bias, const float *__restrict__ R,
                                                  int rmod, float *__restrict__ C, int M, int N, int K)
{
    constexpr int BK = 32, LDP = BK + 4;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A4 = BM * (BK / 4) / 256, B4 = BN * (BK / 4) / 256;   // float4 loads per thread per stage
    extern __shared__ float4 gsm4[];
    float *sm = reinterpret_cast<float *>(gsm4);
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int wm = (wv >> 1) * (BM / 2), wn = (wv & 1) * (BN / 2);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // global -> register staging: chunk i of this thread = float4 (row = (i*256 + tid) / 8, col4 = (..) % 8).
    // (Plain macros, not lambdas: arrays captured by reference end up in scratch memory.)
    f32x4 ra[A4], rb[B4];
    const float *ga[A4], *gb[B4];
    int so_a[A4], so_b[B4];
#pragma unroll
    for (int i = 0; i < A4; ++i) {
        const int ch = i * 256 + tid, row = ch >> 3, c4 = ch & 7;
        ga[i] = A + min(m0 + row, (int64_t)M - 1) * K + 4 * c4;
        so_a[i] = row * LDP + 4 * c4;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
        const int ch = i * 256 + tid, row = ch >> 3, c4 = ch & 7;
        gb[i] = W + min(n0 + row, (int64_t)N - 1) * K + 4 * c4;
        so_b[i] = BM * LDP + row * LDP + 4 * c4;
    }
    constexpr int STAGE = (BM + BN) * LDP;   // floats per stage: [A tile | B tile]
#pragma unroll
    for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ga[i]);
#pragma unroll
    for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(gb[i]);
#pragma unroll
    for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(sm + so_a[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B4; ++i) *reinterpret_cast<f32x4 *>(sm + so_b[i]) = rb[i];
    __syncthreads();
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        float *cur = sm + (kt & 1) * STAGE;
        float *nxt = sm + ((kt & 1) ^ 1) * STAGE;
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = (kt + 1) * BK;
#pragma unroll
            for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ga[i] + k0);
#pragma unroll
            for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(gb[i] + k0);
        }
        const float *as = cur + (wm + r) * LDP + 4 * h;
        const float *bs = cur + BM * LDP + (wn + r) * LDP + 4 * h;
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = *reinterpret_cast<const f32x4 *>(as + a * 32 * LDP + 8 * c);
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = *reinterpret_cast<const f32x4 *>(bs + b * 32 * LDP + 8 * c);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].z, bv[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].w, bv[b].w, acc[a][b], 0, 0, 0);
                }
        }
        if (more) {   // the other stage was last read before the previous barrier
#pragma unroll
            for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(nxt + so_a[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B4; ++i) *reinterpret_cast<f32x4 *>(nxt + so_b[i]) = rb[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int64_t col = n0 + wn + b * 32 + r;
            if (col >= N) continue;
            const float bsv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + wm + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= M) continue;
                float v = acc[a][b][e] + bsv;
                if (EPI == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if (EPI == EPI_ADD_ROW) v += R[row * N + col];
                if (EPI == EPI_ADD_BCAST) v += R[(int64_t)((uint32_t)row % (uint32_t)rmod) * N + col];   // rows < 2^31 (host check)
                C[row * N + col] = v;
            }
        }
}

// Row-panel variant: block tile 128 x 96, 4 waves stacked along M, wave tile 32 x 96 = three independent 32 x 32
// accumulators (enough to keep the matrix pipe issuing back to back from ONE wave per SIMD).  The head's products
// have M = 8192 rows and N in {384, 768, 1536}: M*N/1024 MFMA tiles = 3, 6, 12 per SIMD of the chip, so a 128 x 96
// block tile gives exactly 256 / 512 / 1024 workgroups -- whole multiples of the 256 CUs, where 128 x 64 tiles
// left a quarter of the chip idle.  K step BK (32 or 64), two LDS stages, same k <-> (step, lane half) bijection
// and the same padded conflict-free rows as k_gemm_lds.
template <int BK, int EPI>
// Split K (gridDim.z > 1, EPI_NONE only): slice z covers k in [z*kchunk, (z+1)*kchunk) and writes its partial
// product to C + z*M*N (bias in slice 0); the consumer (k_layernorm) adds the slices in index order.
__global__ __launch_bounds__(256, 2) void k_gemm_panel(const float *__restrict__ A, const float *__restrict__ W,
                                                    const float *__restrict__ bias, const float *__restrict__ R,
                                                    int rmod, float *__restrict__ C, int M, int N, int K, int kchunk)
{
    constexpr int BM = 128, BN = 96, LDP = BK + 4;
    {
        const int z = blockIdx.z;
        A += (size_t)z * kchunk;
        W += (size_t)z * kchunk;
        C += (size_t)z * M * N;
        if (z) bias = nullptr;
    }
    constexpr int A4 = BM * (BK / 4) / 256, B4 = BN * (BK / 4) / 256;   // float4 loads per thread per stage
    constexpr int CPR = BK / 4;                                         // float4 chunks per row
    extern __shared__ float4 gsm4[];
    float *sm = reinterpret_cast<float *>(gsm4);
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int r = lane & 31, h = lane >> 5;
    // Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2: give XCD k the k-th contiguous
    // eighth of the tile list (column panel fastest), so the panels that share a row block of A -- and the row
    // blocks that share W -- meet in one L2 instead of being fetched once per XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if (total % 8 == 0) {
            const int lin = blockIdx.y * gridDim.x + blockIdx.x;
            const int tile = (lin % 8) * (total / 8) + lin / 8;
            by = tile / gridDim.x;
            bx = tile - by * gridDim.x;
        }
    }
    const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;

    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    f32x4 ra[A4], rb[B4];
    const float *ga[A4], *gb[B4];
    int so_a[A4], so_b[B4];
#pragma unroll
    for (int i = 0; i < A4; ++i) {
        const int ch = i * 256 + tid, row = ch / CPR, c4 = ch % CPR;
        ga[i] = A + min(m0 + row, (int64_t)M - 1) * K + 4 * c4;
        so_a[i] = row * LDP + 4 * c4;
    }
#pragma unroll
    for (int i = 0; i < B4; ++i) {
        const int ch = i * 256 + tid, row = ch / CPR, c4 = ch % CPR;
        gb[i] = W + min(n0 + row, (int64_t)N - 1) * K + 4 * c4;
        so_b[i] = BM * LDP + row * LDP + 4 * c4;
    }
    constexpr int STAGE = (BM + BN) * LDP;
#pragma unroll
    for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ga[i]);
#pragma unroll
    for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(gb[i]);
#pragma unroll
    for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(sm + so_a[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B4; ++i) *reinterpret_cast<f32x4 *>(sm + so_b[i]) = rb[i];
    const int nk = kchunk / BK;
    // Staging pipeline (one register set): tile t+1 travels global -> registers during tile t-1's MFMAs, is written to
    // the free LDS stage at the START of tile t (that stage was last read before the barrier that ended tile t-1),
    // and the registers are re-issued for tile t+2 at once -- every load has a whole tile of MFMAs to land, and the
    // LDS writes sit in the shadow of the previous tile's last MFMAs instead of in front of the barrier.
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ga[i] + BK);
#pragma unroll
        for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(gb[i] + BK);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        float *cur = sm + (kt & 1) * STAGE;
        float *nxt = sm + ((kt & 1) ^ 1) * STAGE;
        if (kt + 1 < nk) {
#pragma unroll
            for (int i = 0; i < A4; ++i) *reinterpret_cast<f32x4 *>(nxt + so_a[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B4; ++i) *reinterpret_cast<f32x4 *>(nxt + so_b[i]) = rb[i];
        }
        if (kt + 2 < nk) {
            const int k0 = (kt + 2) * BK;
#pragma unroll
            for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const f32x4 *>(ga[i] + k0);
#pragma unroll
            for (int i = 0; i < B4; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(gb[i] + k0);
        }
        const float *as = cur + (wv * 32 + r) * LDP + 4 * h;
        const float *bs = cur + BM * LDP + r * LDP + 4 * h;
        // fragments of chunk c+1 are read from LDS before the MFMAs of chunk c are issued
        f32x4 av_n = *reinterpret_cast<const f32x4 *>(as), bv_n[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) bv_n[b] = *reinterpret_cast<const f32x4 *>(bs + b * 32 * LDP);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const f32x4 av = av_n;
            f32x4 bv[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) bv[b] = bv_n[b];
            if (c + 1 < BK / 8) {
                av_n = *reinterpret_cast<const f32x4 *>(as + 8 * (c + 1));
#pragma unroll
                for (int b = 0; b < 3; ++b) bv_n[b] = *reinterpret_cast<const f32x4 *>(bs + b * 32 * LDP + 8 * (c + 1));
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
            // k-major over the three accumulators: consecutive MFMAs are independent
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[b].x, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[b].y, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[b].z, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[b].w, acc[b], 0, 0, 0);
        }
        __syncthreads();
    }
    if (m0 + BM <= M && n0 + BN <= N) {
        // interior tile: no bounds checks, so all residual loads of an accumulator are in flight together
        // (the guarded form below waits for each load before it issues the next)
        const int64_t row0 = m0 + wv * 32 + 4 * h;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int64_t col = n0 + b * 32 + r;
            const float bsv = bias ? bias[col] : 0.f;
            float res[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                if (EPI == EPI_ADD_ROW) res[e] = R[row * N + col];
                else if (EPI == EPI_ADD_BCAST) res[e] = R[(int64_t)((uint32_t)row % (uint32_t)rmod) * N + col];
                else res[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                float v = acc[b][e] + bsv;
                if (EPI == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                if (EPI == EPI_ADD_ROW || EPI == EPI_ADD_BCAST) v += res[e];
                C[row * N + col] = v;
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int64_t col = n0 + b * 32 + r;
        if (col >= N) continue;
        const float bsv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = m0 + wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row >= M) continue;
            float v = acc[b][e] + bsv;
            if (EPI == EPI_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            if (EPI == EPI_ADD_ROW) v += R[row * N + col];
            if (EPI == EPI_ADD_BCAST) v += R[(int64_t)((uint32_t)row % (uint32_t)rmod) * N + col];   // rows < 2^31 (host check)
            C[row * N + col] = v;
        }
    }
}

// Softmax over the S band tokens and the context vectors, one workgroup per sample.
//   Qp  [Nq][E]        projected queries (batch-invariant)
//   KV  [S*B][2E]      row s*B + b = (K | V) of token s of sample b
//   ctx [B*Nq][E]
__global__ __launch_bounds__(256) void k_attn_core(const float *__restrict__ Qp,
                                                   const float *__restrict__ KV,
                                                   float *__restrict__ ctx, int B, int E, int heads,
                                                   int Nq, int S)
{
    // LDS: kv[S][2E + 4] (the sample's K | V rows, fetched once with coalesced 16-byte loads) | q[Nq][E] | P[Nq][heads][S]
    extern __shared__ float4 sm4[];
    float *kv = reinterpret_cast<float *>(sm4);
    const int KP = 2 * E + 4;
    float *q = kv + (size_t)S * KP;
    float *P = q + (size_t)Nq * E;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int hd = E / heads;
    const float scale = 1.0f / sqrtf((float)hd);
    const int row4 = 2 * E / 4;
    for (int t = tid; t < S * row4; t += blockDim.x) {
        const int s = t / row4, c = t - s * row4;
        *reinterpret_cast<float4 *>(kv + (size_t)s * KP + 4 * c) =
            *reinterpret_cast<const float4 *>(KV + ((size_t)s * B + b) * 2 * E + 4 * c);
    }
    for (int t = tid; t < Nq * E / 4; t += blockDim.x)
        *reinterpret_cast<float4 *>(q + 4 * t) = *reinterpret_cast<const float4 *>(Qp + 4 * t);
    __syncthreads();
    const int ndots = Nq * heads * S;
    for (int t = tid; t < ndots; t += blockDim.x) {
        const int s = t % S, hh = (t / S) % heads, i = t / (S * heads);
        const float *qv = q + (size_t)i * E + hh * hd;
        const float *kr = kv + (size_t)s * KP + hh * hd;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(qv[d], kr[d], acc);
        P[t] = acc * scale;
    }
    __syncthreads();
    for (int t = tid; t < Nq * heads; t += blockDim.x) {
        float *p = P + (size_t)t * S;
        float mx = p[0];
        for (int s = 1; s < S; ++s) mx = fmaxf(mx, p[s]);
        float sum = 0.f;
        for (int s = 0; s < S; ++s) {
            p[s] = expf(p[s] - mx);
            sum += p[s];
        }
        for (int s = 0; s < S; ++s) p[s] = p[s] / sum;
    }
    __syncthreads();
    for (int t = tid; t < Nq * E; t += blockDim.x) {
        const int e = t % E, i = t / E, hh = e / hd;
        const float *p = P + ((size_t)i * heads + hh) * S;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc = fmaf(p[s], kv[(size_t)s * KP + E + e], acc);
        ctx[((size_t)b * Nq + i) * E + e] = acc;
    }
}

// y = LayerNorm(x) over rows of length E (one wave per row), optional mean over groups of `pool`
// consecutive rows afterwards is handled by k_mean_rows.
// nparts > 1: x is the sum of nparts [rows][E] slices (split-K partials of the read-out product), added in
// slice order.  Rows of up to 64*NV floats are held in registers between the three passes.
template <int NV>
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ x, const float *__restrict__ w,
                                                   const float *__restrict__ bvec, float *__restrict__ y,
                                                   int64_t rows, int E, float eps, int nparts)
{
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + wave_id();
    if (row >= rows) return;
    const float *xr = x + row * E;
    const size_t pstride = (size_t)rows * E;
    auto value = [&](int e) {
        float a = xr[e];
        for (int q = 1; q < nparts; ++q) a += xr[q * pstride + e];
        return a;
    };
    float val[NV > 0 ? NV : 1];
    float s = 0.f;
    if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = lane + 64 * j;
            val[j] = e < E ? value(e) : 0.f;
            s += val[j];
        }
    } else {
        for (int e = lane; e < E; e += 64) s += value(e);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float mean = s / (float)E;
    float v = 0.f;
    if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float dlt = lane + 64 * j < E ? val[j] - mean : 0.f;
            v = fmaf(dlt, dlt, v);
        }
    } else {
        for (int e = lane; e < E; e += 64) {
            const float dlt = value(e) - mean;
            v = fmaf(dlt, dlt, v);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    const float rstd = 1.0f / sqrtf(v / (float)E + eps);
    if constexpr (NV > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int e = lane + 64 * j;
            if (e < E) y[row * E + e] = (val[j] - mean) * rstd * w[e] + bvec[e];
        }
    } else {
        for (int e = lane; e < E; e += 64) y[row * E + e] = (value(e) - mean) * rstd * w[e] + bvec[e];
    }
}

static void launch_layernorm(const float *x, const float *w, const float *b, float *y, int64_t rows, int E, float eps,
                             int nparts, hipStream_t st)
{
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (E <= 512) hipLaunchKernelGGL(k_layernorm<8>, grid, dim3(256), 0, st, x, w, b, y, rows, E, eps, nparts);
    else if (E <= 1024) hipLaunchKernelGGL(k_layernorm<16>, grid, dim3(256), 0, st, x, w, b, y, rows, E, eps, nparts);
    else hipLaunchKernelGGL(k_layernorm<0>, grid, dim3(256), 0, st, x, w, b, y, rows, E, eps, nparts);
}

// Batch-invariant query projection Qp[n][e] = q[n][:] . W[e][:] + b[e] (Nq <= 64 rows): one wave per output
// element.  A matrix-core tile would be 4/32 full here.
__global__ __launch_bounds__(256) void k_qproj(const float *__restrict__ q, const float *__restrict__ W,
                                               const float *__restrict__ b, float *__restrict__ Qp, int Nq, int E)
{
    const int lane = lane_id(), o = blockIdx.x * 4 + wave_id();
    if (o >= Nq * E) return;
    const int n = o / E, e = o - n * E;
    const float *wr = W + (size_t)e * E, *qr = q + (size_t)n * E;
    float a = 0.f;
    for (int k = lane; k < E; k += 64) a = fmaf(qr[k], wr[k], a);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 0) Qp[o] = a + b[e];
}

__global__ void k_mean_rows(const float *__restrict__ x, float *__restrict__ y, int64_t groups, int n, int E)
{
    const int64_t total = groups * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = i / E;
        const int e = (int)(i - g * E);
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += x[(g * n + j) * E + e];
        y[i] = s / (float)n;
    }
}

// logits = fused @ Wh^T (+bh); BatchNorm1d eval; codes = sign(logits); bit-pack (bit = logit > 0)
__global__ __launch_bounds__(256) void k_hash_tail(const float *__restrict__ fused, int B, int E,
                                                   const float *__restrict__ hw, const float *__restrict__ hb,
                                                   const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                                                   const float *__restrict__ bn_mean,
                                                   const float *__restrict__ bn_var, float eps, int nbits,
                                                   float *__restrict__ logits_out,
                                                   float *__restrict__ codes_out,
                                                   uint64_t *__restrict__ packed_out)
{
    extern __shared__ float xrow[];  // E floats
    const int b = blockIdx.x, tid = threadIdx.x, lane = lane_id();
    for (int e = tid; e < E; e += blockDim.x) xrow[e] = fused[(size_t)b * E + e];
    __syncthreads();
    const int words = (nbits + 63) / 64;
    for (int j0 = 0; j0 < words * 64; j0 += blockDim.x) {
        const int j = j0 + tid;
        float v = 0.f;
        const bool valid = j < nbits;
        if (valid) {
            const float4 *wr = reinterpret_cast<const float4 *>(hw + (size_t)j * E);
            float acc = 0.f;
            for (int e4 = 0; e4 < E / 4; ++e4) {
                const float4 w4 = wr[e4];
                acc = fmaf(w4.x, xrow[4 * e4 + 0], acc);
                acc = fmaf(w4.y, xrow[4 * e4 + 1], acc);
                acc = fmaf(w4.z, xrow[4 * e4 + 2], acc);
                acc = fmaf(w4.w, xrow[4 * e4 + 3], acc);
            }
            v = acc + (hb ? hb[j] : 0.f);
            if (bn_w) v = (v - bn_mean[j]) / sqrtf(bn_var[j] + eps) * bn_w[j] + bn_b[j];
            if (logits_out) logits_out[(size_t)b * nbits + j] = v;
            if (codes_out) codes_out[(size_t)b * nbits + j] = v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v));
        }
        const uint64_t word = __ballot(valid && v > 0.f);
        if (packed_out && lane == 0 && j < words * 64) packed_out[(size_t)b * words + j / 64] = word;
    }
}

// Same arithmetic (one fmaf chain over e per logit, in index order) for 16 samples per workgroup: the 64 weight
// rows of the current code word are staged transposed in LDS (Wt[e][bit], pitch 65), lanes = bits, each wave
// owns 4 samples whose features sit in LDS as float4 per e (one broadcast read).  The per-sample kernel above
// reads every weight row from L2 with a 1.5 KB lane stride and keeps one wave in four busy.
__global__ __launch_bounds__(256) void k_hash_tail16(const float *__restrict__ fused, int B, int E,
                                                     const float *__restrict__ hw, const float *__restrict__ hb,
                                                     const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                                                     const float *__restrict__ bn_mean,
                                                     const float *__restrict__ bn_var, float eps, int nbits,
                                                     float *__restrict__ logits_out,
                                                     float *__restrict__ codes_out,
                                                     uint64_t *__restrict__ packed_out)
{
    extern __shared__ float4 hsm4[];
    float4 *xs4 = hsm4;                                        // [4 waves][E] : x of the wave's 4 samples at e
    float *Wt = reinterpret_cast<float *>(hsm4 + 4 * E);       // [E][65]
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int b0 = blockIdx.x * 16;
    {
        float *xs = reinterpret_cast<float *>(xs4);            // element (wave w, e, sample u) at (w*E + e)*4 + u
#pragma unroll 8
        for (int i = tid; i < 16 * E; i += 256) {              // coalesced along e, all loads independent
            const int sidx = i / E, e = i - sidx * E, b = b0 + sidx;
            xs[((sidx >> 2) * E + e) * 4 + (sidx & 3)] = b < B ? fused[(size_t)b * E + e] : 0.f;
        }
    }
    const int words = (nbits + 63) / 64, E4 = E / 4;
    for (int wd = 0; wd < words; ++wd) {
        __syncthreads();                                       // previous word's Wt fully consumed
        constexpr int UN = 8;                                  // loads in flight per thread
        for (int base = 0; base < 64 * E4; base += 256 * UN) {
            f32x4 w4[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = base + u * 256 + tid, jj = i / E4, e4 = i - jj * E4, j = wd * 64 + jj;
                w4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (i < 64 * E4 && j < nbits) w4[u] = *reinterpret_cast<const f32x4 *>(hw + (size_t)j * E + 4 * e4);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int i = base + u * 256 + tid, jj = i / E4, e4 = i - jj * E4;
                if (i < 64 * E4) {
                    Wt[(4 * e4 + 0) * 65 + jj] = w4[u].x;
                    Wt[(4 * e4 + 1) * 65 + jj] = w4[u].y;
                    Wt[(4 * e4 + 2) * 65 + jj] = w4[u].z;
                    Wt[(4 * e4 + 3) * 65 + jj] = w4[u].w;
                }
            }
        }
        __syncthreads();
        const int j = wd * 64 + lane;
        const bool valid = j < nbits;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const float4 *xw = xs4 + wv * E;
#pragma unroll 16
        for (int e = 0; e < E; ++e) {
            const float w = Wt[e * 65 + lane];
            const float4 x = xw[e];
            a0 = fmaf(w, x.x, a0);
            a1 = fmaf(w, x.y, a1);
            a2 = fmaf(w, x.z, a2);
            a3 = fmaf(w, x.w, a3);
        }
        float scale = 1.f, shift = 0.f, mean = 0.f, bias = 0.f;
        if (valid) {
            bias = hb ? hb[j] : 0.f;
            if (bn_w) { mean = bn_mean[j]; scale = sqrtf(bn_var[j] + eps); shift = bn_b[j]; }
        }
        const float acc[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + 4 * wv + u;
            float v = acc[u] + bias;
            if (bn_w && valid) v = (v - mean) / scale * bn_w[j] + shift;
            const bool on = valid && b < B;
            if (on) {
                if (logits_out) logits_out[(size_t)b * nbits + j] = v;
                if (codes_out) codes_out[(size_t)b * nbits + j] = v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v));
            }
            const uint64_t word = __ballot(on && v > 0.f);
            if (packed_out && lane == 0 && b < B) packed_out[(size_t)b * words + wd] = word;
        }
    }
}

// Matrix-core form of the tail for E % 32 == 0: one workgroup per (32 samples, 32 bits), the contraction cut into four
// k ranges (one per wave; partial tiles summed in wave order through LDS -- a fixed order, so a sample's logits do not
// depend on the batch it arrives in).  Operands go global -> registers in the MFMA layout (lane (r, h): row r, k in
// [8c + 4h, 8c + 4h + 4)), every load of a wave in flight at once.  The 16-samples-per-workgroup VALU kernel above spends
// most of its 18.6 us transposing the hash matrix into LDS in every workgroup.
__global__ __launch_bounds__(256) void k_hash_tail_mfma(const float *__restrict__ fused, int B, int E,
                                                        const float *__restrict__ hw, const float *__restrict__ hb,
                                                        const float *__restrict__ bn_w, const float *__restrict__ bn_b,
                                                        const float *__restrict__ bn_mean,
                                                        const float *__restrict__ bn_var, float eps, int nbits,
                                                        float *__restrict__ logits_out, float *__restrict__ codes_out,
                                                        uint64_t *__restrict__ packed_out)
{
    __shared__ float part[4][16][64];                          // [wave][accumulator element][lane]
    const int lane = lane_id(), wv = wave_id(), r = lane & 31, h = lane >> 5;
    const int b0 = blockIdx.x * 32, n = blockIdx.y;             // samples b0 .. b0+31, bits 32n .. 32n+31
    const int cpw = E / 32;                                     // k-chunks of 8 per wave
    const float *arow = fused + (size_t)min(b0 + r, B - 1) * E + (size_t)wv * cpw * 8 + 4 * h;
    const float *brow = hw + (size_t)min(n * 32 + r, nbits - 1) * E + (size_t)wv * cpw * 8 + 4 * h;
    f32x16 acc[2];                                              // even / odd chunks: two independent chains
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    constexpr int CB = 12;                                      // chunks per batch of loads: all of E = 384 in flight at once
    for (int c0 = 0; c0 < cpw; c0 += CB) {
        f32x4 av[CB], bv[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = min(c0 + u, cpw - 1);
            av[u] = *reinterpret_cast<const f32x4 *>(arow + 8 * c);
            bv[u] = *reinterpret_cast<const f32x4 *>(brow + 8 * c);
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c0 + u < cpw) {                                 // uniform
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc[u & 1], 0, 0, 0);
                acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc[u & 1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[wv][e][lane] = acc[0][e] + acc[1][e];
    __syncthreads();
    // wave w finishes accumulator elements 4w .. 4w+3: sample row 8*(e>>2) + 4h + (e&3), bit 32n + r
    const int words = (nbits + 63) / 64;
    const int j = 32 * n + r;
    const bool valid = j < nbits;
    float bias = 0.f, mean = 0.f, scale = 1.f, gain = 1.f, shift = 0.f;
    if (valid) {
        bias = hb ? hb[j] : 0.f;
        if (bn_w) { mean = bn_mean[j]; scale = sqrtf(bn_var[j] + eps); gain = bn_w[j]; shift = bn_b[j]; }
    }
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
        const int e = 4 * wv + ee;
        const int b = b0 + 8 * (e >> 2) + 4 * h + (e & 3);
        float v = ((part[0][e][lane] + part[1][e][lane]) + part[2][e][lane]) + part[3][e][lane];
        v += bias;
        if (bn_w) v = (v - mean) / scale * gain + shift;
        if (valid && b < B) {
            if (logits_out) logits_out[(size_t)b * nbits + j] = v;
            if (codes_out) codes_out[(size_t)b * nbits + j] = v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v));
        }
        const uint64_t bal = __ballot(valid && v > 0.f);        // low half: the sample of lanes 0-31, high half: of lanes 32-63
        if (packed_out && r == 0 && b < B)                      // this workgroup's 32 bits = one half of a packed word
            reinterpret_cast<uint32_t *>(packed_out)[((size_t)b * words + (n >> 1)) * 2 + (n & 1)] = (uint32_t)(bal >> (32 * h));
    }
    // an odd number of 32-bit blocks: the upper half of the last word is nobody's -- zero it
    if (packed_out && n == (int)gridDim.y - 1 && (gridDim.y & 1) && wv == 0 && lane < 32 && b0 + lane < B)
        reinterpret_cast<uint32_t *>(packed_out)[((size_t)(b0 + lane) * words + (n >> 1)) * 2 + 1] = 0u;
}

template <int BM, int BN, int EPI>
static void launch_gemm_lds(const float *A, const float *W, const float *bias, const float *R, int rmod, float *C,
                            int M, int N, int K, hipStream_t st)
{
    constexpr size_t lds = (size_t)2 * (BM + BN) * 36 * sizeof(float);
    auto kern = k_gemm_lds<BM, BN, EPI>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)ceil_div(N, BN), (unsigned)ceil_div(M, BM));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, A, W, bias, R, rmod, C, M, N, K);
}

template <int BK, int EPI>
static void launch_gemm_panel(const float *A, const float *W, const float *bias, const float *R, int rmod, float *C,
                              int M, int N, int K, hipStream_t st, int ksplit = 1)
{
    constexpr size_t lds = (size_t)2 * (128 + 96) * (BK + 4) * sizeof(float);
    auto kern = k_gemm_panel<BK, EPI>;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid((unsigned)ceil_div(N, 96), (unsigned)ceil_div(M, 128), (unsigned)ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, A, W, bias, R, rmod, C, M, N, K, K / ksplit);
}

// Slices the read-out product would be cut into (1 = no split): short-M products leave most CUs without a
// panel, so K is cut until about every CU has one.  C must then hold that many [M][N] partials.
static int readout_ksplit(int M, int N, int K)
{
    if (::wv::tune("WV_GEMM") || K % 64 || N % 96 || M < 128) return 1;
    const int64_t panels = ceil_div(M, 128) * ceil_div(N, 96);
    for (int ks = 8; ks >= 2; ks >>= 1)
        if (K % (ks * 64) == 0 && K / ks >= 128 && panels * ks <= 384) return ks;
    return 1;
}

template <int EPI>
static void launch_gemm(const float *A, const float *W, const float *bias, const float *R, int rmod,
                        float *C, int M, int N, int K, hipStream_t st)
{
    const char *force = ::wv::tune("WV_GEMM");   // "stream" / "lds" / "panel32" / "panel64" pin a kernel (tests / tuning)
    const int64_t panels = ceil_div(M, 128) * ceil_div(N, 96);
    const bool panel_ok = K % 64 == 0 && N % 96 == 0 && M >= 128;
    if (panel_ok && force && !strcmp(force, "panel32")) return launch_gemm_panel<32, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
    if (panel_ok && force && !strcmp(force, "panel64")) return launch_gemm_panel<64, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
    if (panel_ok && !force && panels >= 256) {
        // one workgroup per CU: nothing else hides the stage hand-over, so take the long K step
        if (panels < 512) return launch_gemm_panel<64, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
        return launch_gemm_panel<32, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
    }
    // measured on MI355X at the head's shapes: short-K, wide-N products (mlp.0: K=384, N=1536) run faster on
    // the LDS-free kernel (118 vs 142 us); everything else on the LDS-tiled one
    const bool prefer_stream = ((K <= 512 && N >= 1024) && !(force && !strcmp(force, "lds"))) || (force && !strcmp(force, "stream"));
    if (K % 32 == 0 && M >= 64 && !prefer_stream) {
        // largest tile that still gives every CU about two workgroups
        if (ceil_div(M, 128) * ceil_div(N, 128) >= 512) return launch_gemm_lds<128, 128, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
        if (ceil_div(M, 128) * ceil_div(N, 64) >= 384) return launch_gemm_lds<128, 64, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
        return launch_gemm_lds<64, 64, EPI>(A, W, bias, R, rmod, C, M, N, K, st);
    }
    // big tile when it still yields >= 256 workgroups, else the small one
    const int64_t big = ceil_div(M, 128) * ceil_div(N, 128);
    if (big >= 256) {
        dim3 grid((unsigned)ceil_div(N, 128), (unsigned)ceil_div(M, 128));
        hipLaunchKernelGGL((k_gemm_nt<2, 2, EPI>), grid, dim3(256), 0, st, A, W, bias, R, rmod, C, M, N, K);
    } else {
        dim3 grid((unsigned)ceil_div(N, 64), (unsigned)ceil_div(M, 64));
        hipLaunchKernelGGL((k_gemm_nt<1, 1, EPI>), grid, dim3(256), 0, st, A, W, bias, R, rmod, C, M, N, K);
    }
}

struct HeadWs {
    float *Qp, *KV, *ctx, *x1, *x1n, *hid, *x2, *pooled, *pre;
    size_t bytes;
};

static HeadWs carve(const wv_head_params *p, int B, void *base)
{
    const size_t E = p->embed_dim, Nq = p->num_queries, S = p->num_tokens;
    const size_t rows = (size_t)B * Nq;
    size_t off = 0;
    auto take = [&](size_t n) {
        float *r = base ? (float *)((char *)base + off) : nullptr;
        off += align_up((int64_t)(n * sizeof(float)), 256);
        return r;
    };
    HeadWs w;
    w.Qp = take(Nq * E);
    w.KV = take(S * B * 2 * E);
    w.ctx = take(rows * E);
    w.x1 = take(rows * E);
    w.x1n = take(rows * E);
    w.hid = take(rows * 4 * E);
    w.x2 = take(rows * E);
    w.pooled = take((size_t)B * E);
    w.pre = take((size_t)B * E * 8);   // up to 8 split-K partials of the read-out product
    w.bytes = off;
    return w;
}

// head_front.hip: the fused front (band features -> x2 in one launch) and its prepared weight stream
size_t head_front_prepared_bytes(const wv_head_params *p);
int head_front_prepare(const wv_head_params *p, void *prepared, hipStream_t st);
int head_front_launch(const wv_head_params *p, const float *feats, int B, float *x2, int mode, hipStream_t st);

}  // namespace wv

using namespace wv;

static int check_head(const wv_head_params *p, int B)
{
    WV_REQUIRE(p, "band_attn_pool: null params");
    WV_REQUIRE(B >= 0, "band_attn_pool: B=%d", B);
    WV_REQUIRE(p->embed_dim >= 8 && p->embed_dim % 8 == 0, "band_attn_pool: embed_dim=%d must be a multiple of 8",
               p->embed_dim);
    WV_REQUIRE(p->num_heads >= 1 && p->embed_dim % p->num_heads == 0,
               "band_attn_pool: embed_dim %d not divisible by num_heads %d", p->embed_dim, p->num_heads);
    WV_REQUIRE(p->num_queries >= 1 && p->num_queries <= 64, "band_attn_pool: num_queries=%d", p->num_queries);
    WV_REQUIRE(p->num_tokens >= 1 && p->num_tokens <= 64, "band_attn_pool: num_tokens=%d", p->num_tokens);
    WV_REQUIRE(p->q_eff && p->in_proj_w && p->in_proj_b && p->attn_out_w && p->attn_out_b && p->norm1_w &&
                   p->norm1_b && p->mlp0_w && p->mlp0_b && p->mlp2_w && p->mlp2_b && p->out_w && p->out_b &&
                   p->norm2_w && p->norm2_b,
               "band_attn_pool: null weight pointer");
    return WV_OK;
}

extern "C" size_t wv_band_attn_pool_workspace_bytes(const wv_head_params *p, int B)
{
    if (!p || B <= 0) return 0;
    return carve(p, B, nullptr).bytes;
}

extern "C" int wv_band_attn_qproj(const wv_head_params *p, float *q_proj_out, void *stream)
{
    int rc = check_head(p, 1);
    if (rc) return rc;
    WV_REQUIRE(q_proj_out, "band_attn_qproj: null buffer");
    hipLaunchKernelGGL(k_qproj, dim3((unsigned)ceil_div(p->num_queries * p->embed_dim, 4)), dim3(256), 0, (hipStream_t)stream,
                       p->q_eff, p->in_proj_w, p->in_proj_b, q_proj_out, p->num_queries, p->embed_dim);
    WV_CHECK_LAUNCH("k_qproj");
    return WV_OK;
}

extern "C" size_t wv_band_attn_prepared_bytes(const wv_head_params *p)
{
    if (!p || check_head(p, 1)) return 0;
    return head_front_prepared_bytes(p);
}

extern "C" int wv_band_attn_prepare(const wv_head_params *p, void *prepared_out, void *stream)
{
    int rc = check_head(p, 1);
    if (rc) return rc;
    WV_REQUIRE(prepared_out, "band_attn_prepare: null buffer");
    return head_front_prepare(p, prepared_out, (hipStream_t)stream);
}

extern "C" int wv_band_attn_pool(const wv_head_params *p, const float *feats, int B, float *out,
                                 void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_head(p, B);
    if (rc) return rc;
    WV_REQUIRE(feats && out, "band_attn_pool: null buffer");
    WV_REQUIRE((int64_t)B * std::max(p->num_queries, p->num_tokens) < (1ll << 31), "band_attn_pool: B=%d too large", B);
    if (B == 0) return WV_OK;
    const size_t need = carve(p, B, nullptr).bytes;
    if (!workspace || workspace_bytes < need)
        WV_FAIL(WV_ENOMEM, "band_attn_pool: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int E = p->embed_dim, Nq = p->num_queries, S = p->num_tokens, rows = B * Nq;
    HeadWs w = carve(p, B, workspace);

    // prepared weights: everything up to x2 in one launch when the batch fills the chip (WV_HEAD_FRONT=0 / 1 pins
    // the separate launches / the one-launch front for tests and A/B runs)
    const char *pin = ::wv::tune("WV_HEAD_FRONT");
    const int mode = pin && !strcmp(pin, "0") ? 0 : pin && !strcmp(pin, "1") ? 1 : -1;
    const bool fused = p->prepared && head_front_launch(p, feats, B, w.x2, mode, st);
    if (!fused) {
        // Q projection (batch-invariant): Qp = q_eff @ Wq^T + bq -- taken from the caller when it was made ahead of time
        const float *Qp = p->q_proj ? p->q_proj : reinterpret_cast<const float *>(p->prepared);   // the blob starts with it
        if (!Qp) {
            hipLaunchKernelGGL(k_qproj, dim3((unsigned)ceil_div(Nq * E, 4)), dim3(256), 0, st, p->q_eff, p->in_proj_w,
                               p->in_proj_b, w.Qp, Nq, E);
            Qp = w.Qp;
        }
        // K | V projection of all S*B tokens: rows E..3E of in_proj_weight
        launch_gemm<EPI_NONE>(feats, p->in_proj_w + (size_t)E * E, p->in_proj_b + E, nullptr, 1, w.KV, S * B, 2 * E, E, st);
        const size_t sm = ((size_t)S * (2 * E + 4) + (size_t)Nq * E + (size_t)Nq * p->num_heads * S) * sizeof(float);
        if (sm > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_core), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL(k_attn_core, dim3(B), dim3(256), sm, st, Qp, w.KV, w.ctx, B, E, p->num_heads, Nq, S);
        // x1 = q_eff + ctx @ Wo^T + bo ; x1n = LN1(x1)
        launch_gemm<EPI_ADD_BCAST>(w.ctx, p->attn_out_w, p->attn_out_b, p->q_eff, Nq, w.x1, rows, E, E, st);
        launch_layernorm(w.x1, p->norm1_w, p->norm1_b, w.x1n, (int64_t)rows, E, p->ln_eps, 1, st);
        // x2 = x1n + GELU(x1n @ W0^T + b0) @ W2^T + b2
        launch_gemm<EPI_GELU>(w.x1n, p->mlp0_w, p->mlp0_b, nullptr, 1, w.hid, rows, 4 * E, E, st);
        launch_gemm<EPI_ADD_ROW>(w.hid, p->mlp2_w, p->mlp2_b, w.x1n, 1, w.x2, rows, E, 4 * E, st);
    }
    // read-out: concat (a [B][Nq*E] view of x2) or mean over the queries, then Linear + LN2
    const float *ro_in = w.x2;
    int ro_k = Nq * E;
    if (p->pool_mean) {
        hipLaunchKernelGGL(k_mean_rows, dim3((unsigned)std::min<int64_t>(ceil_div((int64_t)B * E, 256), 4096)),
                           dim3(256), 0, st, w.x2, w.pooled, (int64_t)B, Nq, E);
        ro_in = w.pooled;
        ro_k = E;
    }
    const int ks = readout_ksplit(B, E, ro_k);
    if (ks > 1) launch_gemm_panel<64, EPI_NONE>(ro_in, p->out_w, p->out_b, nullptr, 1, w.pre, B, E, ro_k, st, ks);
    else launch_gemm<EPI_NONE>(ro_in, p->out_w, p->out_b, nullptr, 1, w.pre, B, E, ro_k, st);
    launch_layernorm(w.pre, p->norm2_w, p->norm2_b, out, (int64_t)B, E, p->ln_eps, ks, st);
    WV_CHECK_LAUNCH("band_attn_pool");
    return WV_OK;
}

extern "C" int wv_hash_tail(const float *fused, int B, int E, const float *hash_w, const float *hash_b,
                            const float *bn_w, const float *bn_b, const float *bn_mean,
                            const float *bn_var, float bn_eps, int nbits, float *logits_out,
                            float *codes_out, uint64_t *packed_out, void *stream)
{
    WV_REQUIRE(fused && hash_w, "hash_tail: null buffer");
    WV_REQUIRE(B >= 0 && E >= 4 && E % 4 == 0 && nbits >= 1, "hash_tail: bad shape B=%d E=%d nbits=%d", B, E, nbits);
    WV_REQUIRE(!bn_w || (bn_b && bn_mean && bn_var), "hash_tail: incomplete BatchNorm parameters");
    WV_REQUIRE(E * sizeof(float) <= 48 * 1024, "hash_tail: E=%d too large", E);
    if (B == 0) return WV_OK;
    const size_t lds16 = ((size_t)4 * E * 4 + (size_t)E * 65) * sizeof(float);
    const char *pin = ::wv::tune("WV_HASH_TAIL");                   // "simple" / "valu16" / "mfma" pin a kernel (tests, A/B runs)
    const bool simple = ::wv::tune("WV_HASH_TAIL_SIMPLE") || (pin && !strcmp(pin, "simple"));
    if (!simple && !(pin && !strcmp(pin, "valu16")) && E % 32 == 0 && (B >= 32 || (pin && !strcmp(pin, "mfma")))) {
        const dim3 grid((unsigned)ceil_div(B, 32), (unsigned)ceil_div(nbits, 32));
        hipLaunchKernelGGL(k_hash_tail_mfma, grid, dim3(256), 0, (hipStream_t)stream, fused, B, E, hash_w, hash_b, bn_w, bn_b,
                           bn_mean, bn_var, bn_eps, nbits, logits_out, codes_out, packed_out);
    } else if (lds16 <= (size_t)kMaxLdsBytes - 1024 && B >= 64 && !simple) {
        if (lds16 > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_hash_tail16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        hipLaunchKernelGGL(k_hash_tail16, dim3((unsigned)ceil_div(B, 16)), dim3(256), lds16, (hipStream_t)stream, fused, B,
                           E, hash_w, hash_b, bn_w, bn_b, bn_mean, bn_var, bn_eps, nbits, logits_out, codes_out,
                           packed_out);
    } else {
        hipLaunchKernelGGL(k_hash_tail, dim3(B), dim3(256), E * sizeof(float), (hipStream_t)stream, fused, B, E,
                           hash_w, hash_b, bn_w, bn_b, bn_mean, bn_var, bn_eps, nbits, logits_out, codes_out,
                           packed_out);
    }
    WV_CHECK_LAUNCH("k_hash_tail");
    return WV_OK;
}

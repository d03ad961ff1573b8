// Fused front of the band-attention head: band features -> x2 (everything up to the read-out product) in ONE launch.
//
// Reference: CrossAttentionBottleneckHeadAdvanced.forward, /root/reference/main/models/multi_dino_attention.py:1111-1141
// (nn.MultiheadAttention of Nq learned query tokens over the S = 4 band tokens, residual + LayerNorm, GELU MLP with
// residual).  The separate-launch path in head.hip stays as the general fallback; this kernel takes the shapes the
// reference actually runs (E = 384 = DINOv2 ViT-S, 4 band tokens, 4 or 8 queries).
//
// Why one kernel.  Per sample the head is 14 MFLOP of small dense products whose intermediates (K|V 12 KB, ctx, x1,
// hidden 24 KB, ...) used to travel through HBM/L2 between seven launches, each launch with its own ramp and tail.
// Here one workgroup owns 32 rows (8 samples x 4 queries) from the band features to x2:
//   * every intermediate lives in LDS (three 32 x 384 fp32 tiles, 149 KB: one workgroup per CU, one wave per SIMD);
//   * wave w owns output columns [96w, 96w + 96) of every product, so the B operand (weights) is never shared inside
//     the workgroup: it goes global -> registers directly, from a copy of the weights that wv_band_attn_prepare laid
//     out in MFMA-fragment order, in exactly the order the wave consumes it ("stream": 3 KiB per k-chunk of 8, each
//     wave-instruction reads 1 KiB contiguous).  All workgroups read the same 5.9 MB, so it is served from L2.
//     A 4-entry register ring keeps 3 chunks in flight; there is NO barrier inside a product and the ring keeps
//     prefetching across the phase boundaries (raw s_barrier + lgkmcnt only: nothing drains the vector-memory queue);
//   * the A operand is read from the LDS tile (388-float rows: conflict-free 16-byte fragment reads);
//   * the K projection disappears: the query tokens are parameters, so scores = feats . (Wk_h^T q_ih) / sqrt(hd) is one
//     32-column product with a matrix folded at prepare time (the bias term is constant over the tokens and cancels
//     in the softmax).  The 4 token rows of a sample sit in ONE lane of the 32x32 accumulator layout (rows 4h..4h+3
//     of every group of 8), so softmax-weighted mixing of V happens in registers;
//   * hidden activations are produced in four 384-column slices, each consumed at once as a K slice of mlp.2
//     (x2 accumulators stay in registers across the slices), so the 32 x 1536 hidden tile never exists.
// MFMA work per wave: 5,760 + 48 v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate).  Measured structure ceiling
// (tools/stream_mfma_test.hip): 124-126 TFLOP/s = 0.92 of a bare MFMA loop at one wave per SIMD.
#include "common.hpp"

namespace wv {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int HF_E = 384;            // embed dim the fused kernel is built for
constexpr int HF_LDA = HF_E + 4;     // LDS row pitch (floats)
constexpr int HF_KC = HF_E / 8;      // k-chunks of 8 per 384-long contraction
constexpr int HF_D = 4;              // ring depth (entries; 8 measured the same); phase lengths are multiples of it (scores: padded)
constexpr int HF_ENTRY4 = 3 * 64;    // float4 per stream entry: 3 fragments x 64 lanes
constexpr int HF_TILE = 32 * HF_LDA; // floats per LDS tile

struct HeadFrontPlan {
    int ok, nq, heads, nsc, kw, ns;  // nsc: 32-column score blocks, kw: waves sharing one block along K, ns: score entries per wave
    int nsp;                         // ns rounded up to the ring depth (the extra entries are zeros)
    size_t entries;                  // stream entries per wave (without the ring's over-read pad)
    size_t wave_stride4;             // float4 per wave stream (pad included)
    size_t qp_bytes, bytes;
};

static HeadFrontPlan head_front_plan(const wv_head_params *p)
{
    HeadFrontPlan g{};
    if (!p || p->embed_dim != HF_E || p->num_tokens != 4 || (p->num_queries != 4 && p->num_queries != 8)) return g;
    if (p->num_heads < 1 || HF_E % p->num_heads) return g;
    const int cols = p->num_queries * p->num_heads;
    int nsc = (int)ceil_div(cols, 32);
    if (nsc == 3) nsc = 4;
    if (nsc > 4) return g;
    g.nq = p->num_queries;
    g.heads = p->num_heads;
    g.nsc = nsc;
    g.kw = 4 / nsc;
    g.ns = (HF_KC / g.kw) / 3;
    g.nsp = (int)align_up(g.ns, HF_D);
    g.entries = (size_t)g.nsp + 2 * HF_KC + 4 * 2 * HF_KC;
    g.wave_stride4 = (g.entries + HF_D) * HF_ENTRY4;
    g.qp_bytes = (size_t)align_up((int64_t)g.nq * HF_E * sizeof(float), 256);
    g.bytes = g.qp_bytes + 4 * g.wave_stride4 * sizeof(f32x4);
    g.ok = 1;
    return g;
}

size_t head_front_prepared_bytes(const wv_head_params *p) { return head_front_plan(p).bytes; }

// ------------------------------------------------------------------------------------------------ prepare
// One thread per float4 of the four wave streams.  Lane (r, h) of a fragment holds W[n0 + r][8c + 4h .. 8c + 4h + 4):
// the same k <-> (MFMA step, lane half) bijection as the A fragments read from LDS.
__global__ __launch_bounds__(256) void k_head_pack(const float *__restrict__ in_proj_w, const float *__restrict__ Qp,
                                                   const float *__restrict__ Wo, const float *__restrict__ W0,
                                                   const float *__restrict__ W2, f32x4 *__restrict__ stream, int nq,
                                                   int heads, int kw, int ns, int nsp, size_t entries, size_t wave_stride4)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 4 * wave_stride4) return;
    const int w = (int)(idx / wave_stride4);
    const size_t rem = idx - (size_t)w * wave_stride4;
    const size_t entry = rem / HF_ENTRY4;
    const int q = (int)(rem - entry * HF_ENTRY4), slot = q >> 6, lane = q & 63, r = lane & 31, h = lane >> 5;
    constexpr int E = HF_E;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (entry < (size_t)ns) {
        const int sb = w / kw, kpart = w % kw;
        const int chunk = kpart * (HF_KC / kw) + 3 * (int)entry + slot;
        const int sc = sb * 32 + r;
        if (sc < nq * heads) {
            const int i = sc / heads, hh = sc - i * heads, hd = E / heads;
            const float scale = 1.0f / sqrtf((float)hd);
            const float *qv = Qp + (size_t)i * E + hh * hd;
            const float *wk = in_proj_w + (size_t)E * E + (size_t)hh * hd * E + 8 * chunk + 4 * h;
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int d = 0; d < hd; ++d) {
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(wk + (size_t)d * E);
                a[0] = fmaf(qv[d], kv.x, a[0]);
                a[1] = fmaf(qv[d], kv.y, a[1]);
                a[2] = fmaf(qv[d], kv.z, a[2]);
                a[3] = fmaf(qv[d], kv.w, a[3]);
            }
            v = f32x4{a[0] * scale, a[1] * scale, a[2] * scale, a[3] * scale};
        }
    } else if (entry >= (size_t)nsp && entry < entries) {
        size_t m = entry - nsp;
        const int n = 96 * w + 32 * slot + r;
        const float *src;
        if (m < (size_t)HF_KC) src = in_proj_w + (size_t)2 * E * E + (size_t)n * E + 8 * m + 4 * h;
        else if (m < (size_t)2 * HF_KC) src = Wo + (size_t)n * E + 8 * (m - HF_KC) + 4 * h;
        else {
            m -= 2 * HF_KC;
            const int j = (int)(m / (2 * HF_KC)), mm = (int)(m % (2 * HF_KC));
            if (mm < HF_KC) src = W0 + (size_t)(E * j + n) * E + 8 * mm + 4 * h;
            else src = W2 + (size_t)n * (4 * E) + E * j + 8 * (mm - HF_KC) + 4 * h;
        }
        v = *reinterpret_cast<const f32x4 *>(src);
    }
    stream[idx] = v;   // entries past the end (the ring's over-read) are zeros
}

// ------------------------------------------------------------------------------------------------ device helpers
// Diagnostic build only (-DWV_HF_STAMPS, tools/build_variant.sh): shader-clock cycles per phase of wave 0, summed over
// the workgroups, in a device array nothing else reads.
#ifdef WV_HF_STAMPS
__device__ unsigned long long g_hf_stamps[16];
#define HF_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) atomicAdd(&g_hf_stamps[i], now_ - stamp_); stamp_ = now_; \
    __builtin_amdgcn_sched_barrier(0); } while (0)
#define HF_STAMP_INIT unsigned long long stamp_ = __builtin_amdgcn_s_memtime()
#else
#define HF_STAMP(i) do { } while (0)
#define HF_STAMP_INIT do { } while (0)
#endif

// LDS-only workgroup barrier: waits for this wave's LDS traffic, never for the vector-memory queue (the weight
// ring stays in flight across it).  The "memory" clobber keeps the compiler from moving LDS accesses over it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define HF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// GELU(v) = 0.5 v (1 + erf(v / sqrt 2)) (nn.GELU() default, multi_dino_attention.py:1098).  erfc(|x|) by Abramowitz &
// Stegun 7.1.26 (absolute error <= 1.5e-7, i.e. fp32 rounding of an O(1) value), 1 + erf taken as erfc(|x|) on the
// negative side so that nothing cancels: 14 instructions against ~40 of the library erff.  The activation runs with
// the matrix pipe idle: VALU work of the SAME wave does not overlap its MFMAs here (activating the A fragments inside
// the mlp.2 product instead, ~66 VALU instructions spread between the 12 MFMAs of each step, took the exposed 6 %
// away and added 17 % to the product: measured, dropped).  -DWV_HF_EXACT_ERF restores erff.
__device__ __forceinline__ float gelu_erf(float v)
{
#ifdef WV_HF_EXACT_ERF
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
#else
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));   // v_rcp_f32 (1 ulp); the IEEE division is 11 instructions
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_abs = p * t * __expf(-x * x);
    return 0.5f * v * (v >= 0.f ? 2.0f - erfc_abs : erfc_abs);
#endif
}

// acc[b] += A[32 x 8*NE] . B_b over NE stream entries (entry c = k-chunk c of the wave's three 32-column blocks).
// Ring invariant on entry and exit: slots 0..D-2 hold (or have in flight) the next D-1 entries, wq is the address of
// the entry after them.  Each step first refills the slot the previous step consumed, then reads the next A fragment,
// then issues its 12 MFMAs -- pinned in that order, the scheduler would otherwise sink the loads to their uses.
template <int NE>
__device__ __forceinline__ void gemm_stream(f32x16 (&acc)[3], f32x4 (&ring)[HF_D][3], const f32x4 *&wq, const float *arow)
{
    static_assert(NE % HF_D == 0, "phase length must be a multiple of the ring depth");
    f32x4 av_n = *reinterpret_cast<const f32x4 *>(arow);
    for (int c = 0; c < NE; c += HF_D) {
#pragma unroll
        for (int d = 0; d < HF_D; ++d) {
            constexpr int D = HF_D;
            const int slot = (d + D - 1) % D;
            const f32x4 av = av_n;
            const int cn = min(c + d + 1, NE - 1);
            // one refill load per group of MFMAs: each issues in the shadow of the MFMA before it (three loads and the
            // LDS read in a row take longer to issue than the 64 cycles the last MFMA of a step covers)
            ring[slot][0] = wq[(size_t)(d * 3 + 0) * 64];
            av_n = *reinterpret_cast<const f32x4 *>(arow + 8 * cn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = HF_MFMA(av.x, ring[d][b].x, acc[b]);
            __builtin_amdgcn_sched_barrier(0);
            ring[slot][1] = wq[(size_t)(d * 3 + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = HF_MFMA(av.y, ring[d][b].y, acc[b]);
            __builtin_amdgcn_sched_barrier(0);
            ring[slot][2] = wq[(size_t)(d * 3 + 2) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = HF_MFMA(av.z, ring[d][b].z, acc[b]);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = HF_MFMA(av.w, ring[d][b].w, acc[b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        wq += (size_t)HF_D * HF_ENTRY4;
    }
}

// Score entries: the three fragments of an entry are three CONSECUTIVE k-chunks of ONE 32-column block, summed into
// three independent accumulators (added up by the caller).  arow points at the wave's first k-chunk.
template <int NE, int NREAL>
__device__ __forceinline__ void scores_stream(f32x16 (&sacc)[3], f32x4 (&ring)[HF_D][3], const f32x4 *&wq, const float *arow)
{
    static_assert(NE % HF_D == 0, "phase length must be a multiple of the ring depth");
    f32x4 a_n[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) a_n[t] = *reinterpret_cast<const f32x4 *>(arow + 8 * t);
    for (int c = 0; c < NE; c += HF_D) {
#pragma unroll
        for (int d = 0; d < HF_D; ++d) {
            constexpr int D = HF_D;
            const int slot = (d + D - 1) % D;
#pragma unroll
            for (int b = 0; b < 3; ++b) ring[slot][b] = wq[(size_t)(d * 3 + b) * 64];
            f32x4 a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = a_n[t];
            const int cn = min(c + d + 1, NREAL - 1);   // padded entries (zero weights) re-read valid A data
#pragma unroll
            for (int t = 0; t < 3; ++t) a_n[t] = *reinterpret_cast<const f32x4 *>(arow + 8 * (3 * cn + t));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) sacc[t] = HF_MFMA(a[t].x, ring[d][t].x, sacc[t]);
#pragma unroll
            for (int t = 0; t < 3; ++t) sacc[t] = HF_MFMA(a[t].y, ring[d][t].y, sacc[t]);
#pragma unroll
            for (int t = 0; t < 3; ++t) sacc[t] = HF_MFMA(a[t].z, ring[d][t].z, sacc[t]);
#pragma unroll
            for (int t = 0; t < 3; ++t) sacc[t] = HF_MFMA(a[t].w, ring[d][t].w, sacc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        wq += (size_t)HF_D * HF_ENTRY4;
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[3])
{
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
}

// Sums each of 16 values over the 32 lanes that share this lane's half with a transposing butterfly (16 cross-lane
// moves instead of 80): every step halves the number of values a lane carries.  Lane r ends up with the total of
// value index r >> 1 (lanes r and r ^ 1 hold the same one).
__device__ __forceinline__ float half_reduce16(const float (&v)[16], int r)
{
    float a[8], b[4], c[2];
    const bool u4 = r & 16, u3 = r & 8, u2 = r & 4, u1 = r & 2;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = (u4 ? v[k + 8] : v[k]) + __shfl_xor(u4 ? v[k] : v[k + 8], 16, 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = (u3 ? a[k + 4] : a[k]) + __shfl_xor(u3 ? a[k] : a[k + 4], 8, 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) c[k] = (u2 ? b[k + 2] : b[k]) + __shfl_xor(u2 ? b[k] : b[k + 2], 4, 64);
    float d = (u1 ? c[1] : c[0]) + __shfl_xor(u1 ? c[0] : c[1], 2, 64);
    d += __shfl_xor(d, 1, 64);
    return d;
}

// ------------------------------------------------------------------------------------------------ the kernel
// feats [4][B][E]; x2 [B*NQ][E].  Rows of a workgroup: V phase  rr = 4*bl + s  (sample bl of the block, token s);
// afterwards  row = NQ*bl + i  (query i) = global row 32*blockIdx.x + row.
// Accumulator element e of lane (r, h): row 8*(e>>2) + 4*h + (e&3), column 32*b + r of the wave's 96.
template <int NQ, int NSC>
__global__ __launch_bounds__(256, 1) void k_head_front(const float *__restrict__ feats, const f32x4 *__restrict__ stream,
                                                       size_t wave_stride4, const float *__restrict__ bv,
                                                       const float *__restrict__ bo, const float *__restrict__ q_eff,
                                                       const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                       const float *__restrict__ b0, const float *__restrict__ b2,
                                                       float *__restrict__ x2, int B, int heads, float eps)
{
    constexpr int E = HF_E, LDA = HF_LDA, SPB = 32 / NQ, KW = 4 / NSC, NS = (HF_KC / KW) / 3;
    constexpr int SPP = 33;                                   // pitch of a score-partial row
    extern __shared__ float4 hf_sm4[];
    float *T0 = reinterpret_cast<float *>(hf_sm4);            // feats -> ctx -> x1n
    float *H = T0 + HF_TILE;                                  // two hidden-slice tiles; scratch before the MLP
    float *SP = H;                                            // [4 waves][32][SPP] score partials
    float *P = H + 4 * 32 * SPP;                              // [32 rows (bl, i)][heads][4] softmax weights
    float *ST = H + HF_TILE;                                  // [2][4 waves][32] LayerNorm partial sums
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id(), r = lane & 31, h = lane >> 5;
    const int hd = E / heads;
    const int s0 = blockIdx.x * SPB;

    HF_STAMP_INIT;
    f32x4 ring[HF_D][3];
    const f32x4 *wq = stream + (size_t)wv * wave_stride4 + lane;
#pragma unroll
    for (int d = 0; d < HF_D - 1; ++d)
#pragma unroll
        for (int b = 0; b < 3; ++b) ring[d][b] = wq[(size_t)(d * 3 + b) * 64];
    wq += (size_t)(HF_D - 1) * HF_ENTRY4;
    // Per-lane constants of the epilogues, fetched once: a global load in the middle of the kernel would have to wait
    // for the whole weight ring first (vector-memory results return in order).
    float c_bv[3], c_bo[3], c_lw[3], c_lb[3], c_b2[3], c_b0[4][3], c_q[NQ][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int col = 96 * wv + 32 * b + r;
        c_bv[b] = bv[col];
        c_bo[b] = bo[col];
        c_lw[b] = ln_w[col];
        c_lb[b] = ln_b[col];
        c_b2[b] = b2[col];
#pragma unroll
        for (int j = 0; j < 4; ++j) c_b0[j][b] = b0[E * j + col];
#pragma unroll
        for (int i = 0; i < NQ; ++i) c_q[i][b] = q_eff[i * E + col];
    }

    // band features of the block's samples -> T0 (rows of missing samples are zeros)
    for (int i = tid; i < 32 * (E / 4); i += 256) {
        const int rr = i / (E / 4), c4 = i - rr * (E / 4), bl = rr >> 2, s = rr & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bl < SPB && s0 + bl < B) v = *reinterpret_cast<const f32x4 *>(feats + ((size_t)s * B + s0 + bl) * E + 4 * c4);
        *reinterpret_cast<f32x4 *>(T0 + rr * LDA + 4 * c4) = v;
    }
    lds_barrier();
    HF_STAMP(0);
    const float *arow0 = T0 + r * LDA + 4 * h;
    f32x16 acc[3];

    // ---- scores: wave w covers k-chunks [kpart*48/KW, +48/KW) of score block sb
    {
        zero_acc(acc);
        const int kpart = wv % KW;
        scores_stream<(NS + HF_D - 1) / HF_D * HF_D, NS>(acc, ring, wq, arow0 + 8 * kpart * (HF_KC / KW));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * h + (e & 3);
            SP[(wv * 32 + row) * SPP + r] = (acc[0][e] + acc[1][e]) + acc[2][e];
        }
    }
    lds_barrier();
    HF_STAMP(1);
    for (int it = tid; it < SPB * NQ * heads; it += 256) {     // one (sample, query, head) per thread: softmax over the 4 tokens
        const int hh = it % heads, bi = it / heads, bl = bi / NQ, i = bi - bl * NQ;
        const int sc = i * heads + hh, sb = sc >> 5, col = sc & 31;
        float p[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a = SP[((sb * KW) * 32 + 4 * bl + s) * SPP + col];
#pragma unroll
            for (int k = 1; k < KW; ++k) a += SP[((sb * KW + k) * 32 + 4 * bl + s) * SPP + col];
            p[s] = a;
        }
        const float mx = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            p[s] = expf(p[s] - mx);
            sum += p[s];
        }
        *reinterpret_cast<f32x4 *>(P + (size_t)it * 4) = f32x4{p[0] / sum, p[1] / sum, p[2] / sum, p[3] / sum};
    }

    HF_STAMP(2);
    // ---- V = feats . Wv^T (+ bv), then ctx[bl, i, :] = sum_s P[bl, i, head(col), s] * V[bl, s, :] in registers
    zero_acc(acc);
    gemm_stream<HF_KC>(acc, ring, wq, arow0);
    lds_barrier();                                             // P complete, every wave done with the feature tile
    HF_STAMP(3);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int col = 96 * wv + 32 * b + r, hh = col / hd;
        const float bias = c_bv[b];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bl = h + 2 * j;
            if (bl < SPB) {
                const float v0 = acc[b][4 * j] + bias, v1 = acc[b][4 * j + 1] + bias, v2 = acc[b][4 * j + 2] + bias,
                            v3 = acc[b][4 * j + 3] + bias;
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const f32x4 p4 = *reinterpret_cast<const f32x4 *>(P + ((size_t)(bl * NQ + i) * heads + hh) * 4);
                    float c = p4.x * v0;
                    c = fmaf(p4.y, v1, c);
                    c = fmaf(p4.z, v2, c);
                    c = fmaf(p4.w, v3, c);
                    T0[(bl * NQ + i) * LDA + col] = c;
                }
            }
        }
    }
    lds_barrier();
    HF_STAMP(4);

    // ---- x1 = q_eff + ctx . Wo^T + bo ; x1n = LayerNorm(x1) -> T0
    zero_acc(acc);
    gemm_stream<HF_KC>(acc, ring, wq, arow0);
    HF_STAMP(5);
    {
        float rs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) rs[e] = 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {   // query index = row % NQ, rows go 8*(e>>2) + 4h + (e&3)
                const float qv = NQ == 4 ? c_q[e & 3][b] : (h ? c_q[(4 + (e & 3)) % NQ][b] : c_q[e & 3][b]);
                acc[b][e] = acc[b][e] + c_bo[b] + qv;
                rs[e] += acc[b][e];
            }
        const int erow = r >> 1, srow = 8 * (erow >> 2) + 4 * h + (erow & 3);   // the row whose total half_reduce16 leaves here
        {
            const float t = half_reduce16(rs, r);
            if (!(r & 1)) ST[wv * 32 + srow] = t;
        }
        lds_barrier();                                         // also: every wave is done reading ctx from T0
        float mean[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * h + (e & 3);
            mean[e] = (((ST[row] + ST[32 + row]) + ST[64 + row]) + ST[96 + row]) * (1.0f / (float)E);
            rs[e] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[b][e] -= mean[e];
                rs[e] = fmaf(acc[b][e], acc[b][e], rs[e]);
            }
        {
            const float t = half_reduce16(rs, r);
            if (!(r & 1)) ST[128 + wv * 32 + srow] = t;
        }
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * h + (e & 3);
            const float var = (((ST[128 + row] + ST[160 + row]) + ST[192 + row]) + ST[224 + row]) * (1.0f / (float)E);
            mean[e] = 1.0f / sqrtf(var + eps);                 // now the reciprocal standard deviation
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int col = 96 * wv + 32 * b + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 8 * (e >> 2) + 4 * h + (e & 3);
                T0[row * LDA + col] = acc[b][e] * mean[e] * c_lw[b] + c_lb[b];
            }
        }
    }
    lds_barrier();
    HF_STAMP(6);

    // ---- x2 = x1n + GELU(x1n . W0^T + b0) . W2^T + b2, hidden units in four slices of 384
    f32x16 xacc[3];
    zero_acc(xacc);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        zero_acc(acc);
        gemm_stream<HF_KC>(acc, ring, wq, arow0);
        HF_STAMP(7);
        float *Hj = H + (j & 1) * HF_TILE;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int col = 96 * wv + 32 * b + r;
            const float bias = j == 0 ? c_b0[0][b] : j == 1 ? c_b0[1][b] : j == 2 ? c_b0[2][b] : c_b0[3][b];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = 8 * (e >> 2) + 4 * h + (e & 3);
                Hj[row * LDA + col] = gelu_erf(acc[b][e] + bias);
            }
        }
        lds_barrier();   // slice j complete; also orders slice j-1's readers before slice j+1's writers of the same tile
        HF_STAMP(8);
        gemm_stream<HF_KC>(xacc, ring, wq, Hj + r * LDA + 4 * h);
        HF_STAMP(9);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int col = 96 * wv + 32 * b + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * h + (e & 3);
            const int64_t grow = (int64_t)blockIdx.x * 32 + row;
            if (grow < (int64_t)B * NQ) x2[grow * E + col] = xacc[b][e] + c_b2[b] + T0[row * LDA + col];
        }
    }
    HF_STAMP(10);
}

// ------------------------------------------------------------------------------------------------ host side
int head_front_prepare(const wv_head_params *p, void *prepared, hipStream_t st)
{
    const HeadFrontPlan g = head_front_plan(p);
    if (!g.ok) WV_FAIL(WV_ENOTSUP, "band_attn_prepare: no fused kernel for E=%d S=%d Nq=%d heads=%d", p->embed_dim,
                       p->num_tokens, p->num_queries, p->num_heads);
    float *Qp = reinterpret_cast<float *>(prepared);
    int rc = wv_band_attn_qproj(p, Qp, st);
    if (rc) return rc;
    f32x4 *stream = reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(prepared) + g.qp_bytes);
    const size_t total = 4 * g.wave_stride4;
    hipLaunchKernelGGL(k_head_pack, dim3((unsigned)ceil_div((int64_t)total, 256)), dim3(256), 0, st, p->in_proj_w, Qp,
                       p->attn_out_w, p->mlp0_w, p->mlp2_w, stream, g.nq, g.heads, g.kw, g.ns, g.nsp, g.entries, g.wave_stride4);
    WV_CHECK_LAUNCH("k_head_pack");
    return WV_OK;
}

template <int NQ, int NSC>
static void launch_front(const HeadFrontPlan &g, const wv_head_params *p, const float *feats, int B, float *x2, hipStream_t st)
{
    constexpr size_t lds = (size_t)3 * HF_TILE * sizeof(float);
    static_assert(lds <= (size_t)kMaxLdsBytes, "three tiles must fit the CU's LDS");
    auto kern = k_head_front<NQ, NSC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const f32x4 *stream = reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(p->prepared) + g.qp_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(B, 32 / NQ)), dim3(256), lds, st, feats, stream, g.wave_stride4,
                       p->in_proj_b + 2 * HF_E, p->attn_out_b, p->q_eff, p->norm1_w, p->norm1_b, p->mlp0_b, p->mlp2_b, x2, B,
                       p->num_heads, p->ln_eps);
}

// 1 = launched (x2 [B*Nq][E] will hold the MLP output), 0 = separate launches are the better (or only) choice.
// A workgroup runs its 32 rows through all 5,808 MFMAs of a wave whatever the batch: ~215 us even for one sample.  The
// separate launches spread a small batch over the whole chip instead; measured crossover on MI355X (Nq = 4: B = 1024
// -> 241 vs 217 us, B = 1536 -> 252 vs 317 us): from about 9/16 of the 256 CUs on, the one-launch front wins.
// mode: 0 = never, 1 = whenever the configuration has a kernel, -1 = by that rule.
int head_front_launch(const wv_head_params *p, const float *feats, int B, float *x2, int mode, hipStream_t st)
{
    const HeadFrontPlan g = head_front_plan(p);
    if (!g.ok || !p->prepared || mode == 0) return 0;
    if (mode < 0 && ceil_div(B, 32 / g.nq) < 144) return 0;
    if (g.nq == 4 && g.nsc == 1) launch_front<4, 1>(g, p, feats, B, x2, st);
    else if (g.nq == 4 && g.nsc == 2) launch_front<4, 2>(g, p, feats, B, x2, st);
    else if (g.nq == 4 && g.nsc == 4) launch_front<4, 4>(g, p, feats, B, x2, st);
    else if (g.nq == 8 && g.nsc == 1) launch_front<8, 1>(g, p, feats, B, x2, st);
    else if (g.nq == 8 && g.nsc == 2) launch_front<8, 2>(g, p, feats, B, x2, st);
    else if (g.nq == 8 && g.nsc == 4) launch_front<8, 4>(g, p, feats, B, x2, st);
    else return 0;
    return 1;
}

}  // namespace wv

#ifdef WV_HF_STAMPS
// diagnostic build only: read and reset the phase cycle sums (host array of 16)
extern "C" int wv_debug_hf_stamps(unsigned long long *host16)
{
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(wv::g_hf_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -5;
    unsigned long long zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(wv::g_hf_stamps), zero, sizeof(zero)) != hipSuccess) return -5;
    return 0;
}
#endif

// Two-pass SWT kernel for gfx950: all levels of one direction are fused in registers.
//
// The level-n sub-bands are  V_n{lo,hi} V_{n-1}lo .. V_1lo  H_n{lo,hi} H_{n-1}lo .. H_1lo  x :
// the separable a-trous filters of the two axes commute, so instead of ping-ponging every level
// through LDS (k_swt_tiled, swt.hip: 2n passes, 2n-1 barriers) the kernel runs
//   pass H : thread = (row, run of R output columns).  It loads the R + HALO input pixels of its
//            run straight from global memory (wrapped mod W, uint8 -> x/255 exactly), applies
//            levels 1..n along the row IN REGISTERS (in place: level l only reads indices >= i),
//            and writes the two level-n row results (lo, hi) to two LDS planes.
//   pass V : thread = (plane, column).  It reads the TH + HALO values of its column from LDS,
//            applies levels 1..n down the column in registers and streams the 2 x TH outputs
//            (lo, hi -> bands 2*plane, 2*plane+1) to global memory; lanes = adjacent columns, so
//            every store instruction writes one contiguous row segment.
// HALO = (L-1)(2^n-1).  One barrier per tile, LDS = 2 planes of (TH+HALO) x (TW+4) floats
// (49 KB for db2 level 3 -> 3 workgroups per CU), no intermediate level ever leaves registers.
// Per 1-D pass the taps are accumulated in the reference order (m = 0..L-1, fmaf); only the order
// of the (commuting) passes differs from pywt's axis-0-then-axis-1 per level, i.e. fp32 rounding.
#include "common.hpp"
#include "swt_fused.hpp"

namespace wv {

constexpr int kFusedThreads = 256;

template <int L>
struct FTaps {
    float lo[L];
    float hi[L];
};

struct FusedGeom {
    int B, C, H, W;
    int tilesX;
    int in_layout;
    int out_bf16;
};

template <int L, int NLEV, int NOUT>
struct Chain {
    static constexpr int HALO = (L - 1) * ((1 << NLEV) - 1);
    static constexpr int NIN = NOUT + HALO;

    // levels 1 .. NLEV-1 (approximation only), in place: a_l[i] = sum_m lo[m] * a_{l-1}[i + s(L-1-m)]
    template <int LEV>
    static __device__ __forceinline__ void lower(float (&v)[NIN], const float (&lo)[L])
    {
        if constexpr (LEV < NLEV) {
            constexpr int S = 1 << (LEV - 1);
            constexpr int LEN = NIN - (L - 1) * ((1 << LEV) - 1);
#pragma unroll
            for (int i = 0; i < LEN; ++i) {
                float a = lo[0] * v[i + S * (L - 1)];
#pragma unroll
                for (int m = 1; m < L; ++m) a = fmaf(lo[m], v[i + S * (L - 1 - m)], a);
                v[i] = a;
            }
            lower<LEV + 1>(v, lo);
        }
    }
    // level NLEV output i with filter f
    static __device__ __forceinline__ float last(const float (&v)[NIN], const float (&f)[L], int i)
    {
        constexpr int S = 1 << (NLEV - 1);
        float a = f[0] * v[i + S * (L - 1)];
#pragma unroll
        for (int m = 1; m < L; ++m) a = fmaf(f[m], v[i + S * (L - 1 - m)], a);
        return a;
    }
};

__device__ __forceinline__ int wrapi(int v, int n)
{
    while (v < 0) v += n;
    while (v >= n) v -= n;
    return v;
}

// exact fp32 x / 255 for x in 0..255 (verified exhaustively against IEEE division):
// q = x * r ; e = fma(-q, 255, x) ; q' = fma(e, r, q), r = RN(1/255)
__device__ __forceinline__ float u8_to_unit(float x)
{
    const float r = 0.003921568859368563f;  // 0x3b808081
    const float q = x * r;
    const float e = fmaf(-q, 255.0f, x);
    return fmaf(e, r, q);
}

template <int N>
__device__ __forceinline__ float ubyte_f32(uint32_t d)
{
    return (float)((d >> (8 * N)) & 0xffu);  // -> v_cvt_f32_ubyteN
}

// 4 consecutive pixels (x multiple of 4) of channel c at row gy, as fp32 in [0,1]
template <typename InT>
__device__ __forceinline__ float4 load_px4(const InT *__restrict__ in, const FusedGeom &g, int b, int c,
                                           int gy, int gx)
{
    if constexpr (sizeof(InT) == 1) {
        if (g.in_layout == WV_LAYOUT_NCHW) {
            const uint32_t d = *reinterpret_cast<const uint32_t *>(in + (((size_t)b * g.C + c) * g.H + gy) * g.W + gx);
            return make_float4(u8_to_unit(ubyte_f32<0>(d)), u8_to_unit(ubyte_f32<1>(d)),
                               u8_to_unit(ubyte_f32<2>(d)), u8_to_unit(ubyte_f32<3>(d)));
        }
        if (g.C == 3) {  // 4 RGB pixels = 12 bytes = 3 aligned dwords; channel c sits at bytes c, 3+c, 6+c, 9+c
            const uint32_t *p = reinterpret_cast<const uint32_t *>(in + (((size_t)b * g.H + gy) * g.W + gx) * 3);
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            const uint32_t s0 = __builtin_amdgcn_alignbyte(d1, d0, (uint32_t)c);
            const uint32_t s1 = __builtin_amdgcn_alignbyte(d2, d1, (uint32_t)c);
            const uint32_t s2 = __builtin_amdgcn_alignbyte(0u, d2, (uint32_t)c);
            return make_float4(u8_to_unit(ubyte_f32<0>(s0)), u8_to_unit(ubyte_f32<3>(s0)),
                               u8_to_unit(ubyte_f32<2>(s1)), u8_to_unit(ubyte_f32<1>(s2)));
        }
        const InT *p = in + (((size_t)b * g.H + gy) * g.W + gx) * g.C + c;
        return make_float4(u8_to_unit((float)p[0]), u8_to_unit((float)p[g.C]), u8_to_unit((float)p[2 * g.C]),
                           u8_to_unit((float)p[3 * g.C]));
    } else {
        if (g.in_layout == WV_LAYOUT_NCHW)
            return *reinterpret_cast<const float4 *>(in + (((size_t)b * g.C + c) * g.H + gy) * g.W + gx);
        const InT *p = in + (((size_t)b * g.H + gy) * g.W + gx) * g.C + c;
        return make_float4(p[0], p[g.C], p[2 * g.C], p[3 * g.C]);
    }
}

__device__ __forceinline__ void store_out(void *out, size_t off, float v, int bf16)
{
    if (bf16) reinterpret_cast<__hip_bfloat16 *>(out)[off] = __float2bfloat16(v);
    else reinterpret_cast<float *>(out)[off] = v;
}

template <int L, int NLEV, int R, int NRUN, int TH, typename InT>
__global__ __launch_bounds__(kFusedThreads) void k_swt_fused(const InT *__restrict__ in, void *__restrict__ out,
                                                             FusedGeom g, FTaps<L> taps)
{
    using CH = Chain<L, NLEV, R>;
    using CV = Chain<L, NLEV, TH>;
    constexpr int HALO = CH::HALO;
    constexpr int HB = (L / 2 - 1) * ((1 << NLEV) - 1);   // halo before; after = HALO - HB
    constexpr int HBa = (HB + 3) / 4 * 4;
    constexpr int TW = R * NRUN;
    constexpr int RH = TH + HALO;
    constexpr int P = TW + 4;                              // plane pitch (floats)
    constexpr int NG = (HBa - HB + CH::NIN + 3) / 4;       // aligned 4-pixel groups a run loads
    static_assert(R % 4 == 0, "run length must be a multiple of 4");
    __shared__ float4 planes4[2 * RH * P / 4];
    float *planes = reinterpret_cast<float *>(planes4);

    const int tile = blockIdx.x;
    const int ty = tile / g.tilesX, tx = tile - ty * g.tilesX;
    const int c = blockIdx.y, b = blockIdx.z;
    const int x0 = tx * TW, y0 = ty * TH;

    // ------------------------------------------------------------------ pass H
    for (int u = threadIdx.x; u < RH * NRUN; u += kFusedThreads) {
        const int r = u / NRUN, j = u - r * NRUN;
        const int gy = wrapi(y0 - HB + r, g.H);
        const int gx0 = x0 + j * R - HBa;
        float raw[NG * 4];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const float4 p4 = load_px4<InT>(in, g, b, c, gy, wrapi(gx0 + 4 * k, g.W));
            raw[4 * k + 0] = p4.x; raw[4 * k + 1] = p4.y; raw[4 * k + 2] = p4.z; raw[4 * k + 3] = p4.w;
        }
        float v[CH::NIN];
#pragma unroll
        for (int i = 0; i < CH::NIN; ++i) v[i] = raw[i + (HBa - HB)];
        CH::template lower<1>(v, taps.lo);
        float *plo = planes + r * P + j * R;
        float *phi = plo + RH * P;
#pragma unroll
        for (int q4 = 0; q4 < R / 4; ++q4) {
            float4 lo4, hi4;
            lo4.x = CH::last(v, taps.lo, 4 * q4 + 0); hi4.x = CH::last(v, taps.hi, 4 * q4 + 0);
            lo4.y = CH::last(v, taps.lo, 4 * q4 + 1); hi4.y = CH::last(v, taps.hi, 4 * q4 + 1);
            lo4.z = CH::last(v, taps.lo, 4 * q4 + 2); hi4.z = CH::last(v, taps.hi, 4 * q4 + 2);
            lo4.w = CH::last(v, taps.lo, 4 * q4 + 3); hi4.w = CH::last(v, taps.hi, 4 * q4 + 3);
            *reinterpret_cast<float4 *>(plo + 4 * q4) = lo4;
            *reinterpret_cast<float4 *>(phi + 4 * q4) = hi4;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ pass V
    const size_t band = (size_t)g.H * g.W;
    for (int u = threadIdx.x; u < 2 * TW; u += kFusedThreads) {
        const int pl = u / TW, x = u - pl * TW;
        const int gx = x0 + x;
        if (gx >= g.W) continue;
        const float *col = planes + pl * RH * P + x;
        float v[CV::NIN];
#pragma unroll
        for (int i = 0; i < CV::NIN; ++i) v[i] = col[i * P];
        CV::template lower<1>(v, taps.lo);
        const size_t base = (((size_t)b * g.C + c) * 4 + 2 * pl) * band + (size_t)y0 * g.W + gx;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            if (y0 + i < g.H) {
                store_out(out, base + (size_t)i * g.W, CV::last(v, taps.lo, i), g.out_bf16);
                store_out(out, base + band + (size_t)i * g.W, CV::last(v, taps.hi, i), g.out_bf16);
            }
        }
    }
}

template <int L, int NLEV, int R, int NRUN, int TH, typename InT>
static int launch_fused(const void *in, void *out, FusedGeom g, const float *lo, const float *hi, hipStream_t st)
{
    FTaps<L> taps;
    for (int i = 0; i < L; ++i) { taps.lo[i] = lo[i]; taps.hi[i] = hi[i]; }
    constexpr int TW = R * NRUN;
    g.tilesX = (int)ceil_div(g.W, TW);
    dim3 grid((unsigned)(g.tilesX * ceil_div(g.H, TH)), (unsigned)g.C, (unsigned)g.B);
    hipLaunchKernelGGL((k_swt_fused<L, NLEV, R, NRUN, TH, InT>), grid, dim3(kFusedThreads), 0, st,
                       (const InT *)in, out, g, taps);
    WV_CHECK_LAUNCH("k_swt_fused");
    return WV_OK;
}

// tile shapes: wide (TW = 112, for W = 224 and multiples) and narrow (TW = 64)
template <int L, int NLEV, typename InT>
static int pick_shape(const void *in, void *out, const FusedGeom &g, const float *lo, const float *hi,
                      hipStream_t st)
{
    const int64_t waste112 = ceil_div(g.W, 112) * 112 - g.W, waste64 = ceil_div(g.W, 64) * 64 - g.W;
    if (waste112 * 64 <= waste64 * 112)   // compare relative padding
        return launch_fused<L, NLEV, 28, 4, 32, InT>(in, out, g, lo, hi, st);
    return launch_fused<L, NLEV, 16, 4, 32, InT>(in, out, g, lo, hi, st);
}

template <typename InT>
static int pick_filter(int L, int n, const void *in, void *out, const FusedGeom &g, const float *lo,
                       const float *hi, hipStream_t st)
{
#define WV_CASE(LL, NN) \
    if (L == LL && n == NN) return pick_shape<LL, NN, InT>(in, out, g, lo, hi, st)
    WV_CASE(2, 1); WV_CASE(2, 2); WV_CASE(2, 3);
    WV_CASE(4, 1); WV_CASE(4, 2); WV_CASE(4, 3);
    WV_CASE(8, 1);
    WV_CASE(10, 1);
#undef WV_CASE
    return 1;  // not covered
}

bool swt_fused_covers(int L, int n, int W)
{
    const bool cfg = (L == 2 && n <= 3) || (L == 4 && n <= 3) || (L == 8 && n == 1) || (L == 10 && n == 1);
    return cfg && n >= 1 && (W % 4) == 0;
}

int swt_fused_launch(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B, int C, int H,
                     int W, int n, const float *lo, const float *hi, int L, hipStream_t st)
{
    FusedGeom g{};
    g.B = B; g.C = C; g.H = H; g.W = W; g.in_layout = in_layout; g.out_bf16 = out_dtype == WV_DT_BF16;
    if (in_dtype == WV_DT_U8) return pick_filter<uint8_t>(L, n, in, out, g, lo, hi, st);
    return pick_filter<float>(L, n, in, out, g, lo, hi, st);
}

}  // namespace wv

// Sliding-window SWT kernel for gfx950 (full-width rows, persistent over image planes).
//
// Same register-fused two-pass scheme as swt_fused.hip (all levels of one direction in registers;
// the separable a-trous passes of the two axes commute), restructured for the way the output lies
// in memory: out[b][c][band] is one contiguous H x W block, so a workgroup that owns a whole
// (image, channel) plane and walks down it TH rows at a time writes four pure streams.
//   ring   : LDS ring of RH = TH + HALO rows x 2 planes (row-filtered lo / hi), pitch W + 4
//   step s : pass H on the TH new input rows (each input row is row-filtered exactly once: no
//            vertical halo recompute), barrier, pass V on the RH rows now in the ring -> TH output
//            rows x 4 bands, barrier.  The raw pixels of the next step are fetched into registers
//            before pass V, so global-load latency hides behind the column arithmetic.
//   pass H : thread = (row, run of R columns), pixels wrapped mod W / mod H at load time
//   pass V : thread = (plane, column); lanes = adjacent columns -> each store instruction writes one
//            contiguous row segment, consecutive rows follow each other in memory
// uint8 input is converted exactly (x/255 via fma refinement, verified for all 256 values).
#include "common.hpp"
#include "swt_fused.hpp"
#include <type_traits>

namespace wv {

template <int L>
struct STaps {
    float lo[L];
    float hi[L];
};

struct SlideGeom {
    int B, C, H, W;
    int P;          // ring pitch in floats
    int nrun;       // runs per row = ceil(W / R)
    int in_layout;
    int out_bf16;
};

template <int L, int NLEV, int NOUT>
struct SChain {
    static constexpr int HALO = (L - 1) * ((1 << NLEV) - 1);
    static constexpr int NIN = NOUT + HALO;
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
    static __device__ __forceinline__ float last(const float (&v)[NIN], const float (&f)[L], int i)
    {
        constexpr int S = 1 << (NLEV - 1);
        float a = f[0] * v[i + S * (L - 1)];
#pragma unroll
        for (int m = 1; m < L; ++m) a = fmaf(f[m], v[i + S * (L - 1 - m)], a);
        return a;
    }
};

// branch-free wrap, valid for -n <= v < 2n (guaranteed by swt_slide_covers)
__device__ __forceinline__ int swrap1(int v, int n)
{
    v = v < 0 ? v + n : v;
    return v >= n ? v - n : v;
}

__device__ __forceinline__ float s_u8_to_unit(float x)
{
    const float r = 0.003921568859368563f;  // RN(1/255) = 0x3b808081
    const float q = x * r;
    const float e = fmaf(-q, 255.0f, x);
    return fmaf(e, r, q);                   // == RN(x / 255) for every x in 0..255
}

template <int N>
__device__ __forceinline__ float s_ubyte(uint32_t d)
{
    return (float)((d >> (8 * N)) & 0xffu);
}

// LAYOUT: 0 = NCHW, 1 = NHWC with C == 3.  One aligned 4-pixel group of channel c, raw.
template <typename InT, int LAYOUT>
struct SRaw {
    uint32_t d[sizeof(InT) == 1 ? (LAYOUT == 1 ? 3 : 1) : 4];
};

// `img` = uniform base of the (b) image [NHWC] or of the (b, c) plane [NCHW]; off = lane offset in elements
template <typename InT, int LAYOUT>
__device__ __forceinline__ SRaw<InT, LAYOUT> s_fetch4(const InT *__restrict__ img, uint32_t pix, int c)
{
    SRaw<InT, LAYOUT> r;
    if constexpr (sizeof(InT) == 1) {
        if constexpr (LAYOUT == 0) {
            r.d[0] = *reinterpret_cast<const uint32_t *>(img + pix);
        } else {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(img + pix * 3u);
            r.d[0] = p[0]; r.d[1] = p[1]; r.d[2] = p[2];
        }
    } else {
        float4 v;
        if constexpr (LAYOUT == 0) {
            v = *reinterpret_cast<const float4 *>(img + pix);
        } else {
            const InT *p = img + pix * 3u + c;
            v = make_float4(p[0], p[3], p[6], p[9]);
        }
        r.d[0] = __float_as_uint(v.x); r.d[1] = __float_as_uint(v.y);
        r.d[2] = __float_as_uint(v.z); r.d[3] = __float_as_uint(v.w);
    }
    return r;
}

template <typename InT, int LAYOUT>
__device__ __forceinline__ float4 s_convert4(const SRaw<InT, LAYOUT> &r, int c)
{
    if constexpr (sizeof(InT) == 1) {
        if constexpr (LAYOUT == 1) {  // channel c at bytes c, 3+c, 6+c, 9+c of the 12
            const uint32_t s0 = __builtin_amdgcn_alignbyte(r.d[1], r.d[0], (uint32_t)c);
            const uint32_t s1 = __builtin_amdgcn_alignbyte(r.d[2], r.d[1], (uint32_t)c);
            const uint32_t s2 = __builtin_amdgcn_alignbyte(0u, r.d[2], (uint32_t)c);
            return make_float4(s_u8_to_unit(s_ubyte<0>(s0)), s_u8_to_unit(s_ubyte<3>(s0)),
                               s_u8_to_unit(s_ubyte<2>(s1)), s_u8_to_unit(s_ubyte<1>(s2)));
        } else {
            const uint32_t d = r.d[0];
            return make_float4(s_u8_to_unit(s_ubyte<0>(d)), s_u8_to_unit(s_ubyte<1>(d)),
                               s_u8_to_unit(s_ubyte<2>(d)), s_u8_to_unit(s_ubyte<3>(d)));
        }
    } else {
        return make_float4(__uint_as_float(r.d[0]), __uint_as_float(r.d[1]), __uint_as_float(r.d[2]),
                           __uint_as_float(r.d[3]));
    }
}

template <int L, int NLEV, int R, int TH, int NT, typename InT, int LAYOUT, bool BF16>
__global__ __launch_bounds__(NT) void k_swt_slide(const InT *__restrict__ in, void *__restrict__ out,
                                                  SlideGeom g, STaps<L> taps)
{
    using CH = SChain<L, NLEV, R>;
    using CV = SChain<L, NLEV, TH>;
    using Raw = SRaw<InT, LAYOUT>;
    using OutT = typename std::conditional<BF16, __hip_bfloat16, float>::type;
    constexpr int HALO = CH::HALO;
    constexpr int HB = (L / 2 - 1) * ((1 << NLEV) - 1);
    constexpr int HA = HALO - HB;
    constexpr int HBa = (HB + 3) / 4 * 4;
    constexpr int RH = TH + HALO;
    constexpr int NG = (HBa - HB + CH::NIN + 3) / 4;      // 4-pixel groups one run loads
    static_assert(R % 4 == 0, "run length must be a multiple of 4");
    extern __shared__ float4 ring4[];
    float *ring = reinterpret_cast<float *>(ring4);       // [2][RH][P]
    const int P = g.P, W = g.W, H = g.H;
    const int plane_sz = RH * P;
    const int nsteps = (H + TH - 1) / TH;
    const uint32_t band = (uint32_t)H * W;
    const int nplanes = g.B * g.C;
    const int units = TH * g.nrun;                        // <= NT (checked on the host)
    // this thread's (row-in-step, run) for pass H
    const int h_rr = threadIdx.x / g.nrun, h_j = threadIdx.x - h_rr * g.nrun;
    const bool h_active = (int)threadIdx.x < units;

    auto fetch_unit = [&](Raw (&raw)[NG], const InT *img, int c, int y, int j) {
        const uint32_t row = (uint32_t)swrap1(y, H) * (uint32_t)W;
        const int gx0 = j * R - HBa;
#pragma unroll
        for (int k = 0; k < NG; ++k) raw[k] = s_fetch4<InT, LAYOUT>(img, row + (uint32_t)swrap1(gx0 + 4 * k, W), c);
    };
    auto hpass_unit = [&](const Raw (&raw)[NG], int c, int slot, int j) {
        float px[NG * 4];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const float4 p4 = s_convert4<InT, LAYOUT>(raw[k], c);
            px[4 * k + 0] = p4.x; px[4 * k + 1] = p4.y; px[4 * k + 2] = p4.z; px[4 * k + 3] = p4.w;
        }
        float v[CH::NIN];
#pragma unroll
        for (int i = 0; i < CH::NIN; ++i) v[i] = px[i + (HBa - HB)];
        CH::template lower<1>(v, taps.lo);
        float *plo = ring + slot * P + j * R;
        float *phi = plo + plane_sz;
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
    };

    for (int pc = blockIdx.x; pc < nplanes; pc += gridDim.x) {
        const int b = pc / g.C, c = pc - b * g.C;
        // uniform base of the pixels this plane reads
        const InT *img = LAYOUT == 0 ? in + (size_t)pc * band : in + (size_t)b * band * 3;
        OutT *oplane = reinterpret_cast<OutT *>(out) + (size_t)pc * 4 * band;
        // ---- prologue: the HALO rows around the first block (y in [-HB, HA)) -> slots 0 .. HALO-1
        for (int u = threadIdx.x; u < HALO * g.nrun; u += NT) {
            const int rr = u / g.nrun, j = u - rr * g.nrun;
            Raw raw[NG];
            fetch_unit(raw, img, c, rr - HB, j);
            hpass_unit(raw, c, rr, j);
        }
        Raw pre[NG];
        if (h_active) fetch_unit(pre, img, c, HA + h_rr, h_j);
        int slot_new = HALO;     // ring slot of the first new row of the step ((y0 + HALO) % RH)
        int s0 = 0;              // ring slot of virtual row y0 (= y0 % RH)
        for (int s = 0; s < nsteps; ++s) {
            const int y0 = s * TH;
            // ---- pass H on the TH new rows y0 + HA .. y0 + HA + TH - 1 (prefetched)
            if (h_active) {
                int slot = slot_new + h_rr;
                slot = slot >= RH ? slot - RH : slot;
                hpass_unit(pre, c, slot, h_j);
            }
            __syncthreads();
            // ---- prefetch the raw pixels of the next step (hidden behind pass V)
            if (h_active && s + 1 < nsteps) fetch_unit(pre, img, c, y0 + TH + HA + h_rr, h_j);
            // ---- pass V: ring rows s0 .. s0 + RH - 1 (mod RH) -> output rows y0 .. y0 + TH - 1
            OutT *orow0 = oplane + (size_t)y0 * W;   // uniform
            for (int u = threadIdx.x; u < 2 * W; u += NT) {
                const int pl = u >= W ? 1 : 0, x = u - pl * W;
                const float *col = ring + pl * plane_sz + x;
                float v[CV::NIN];
#pragma unroll
                for (int i = 0; i < CV::NIN; ++i) {
                    int slot = s0 + i;
                    slot = slot >= RH ? slot - RH : slot;
                    v[i] = col[slot * P];
                }
                CV::template lower<1>(v, taps.lo);
                const uint32_t lane_off = (uint32_t)(2 * pl) * band + (uint32_t)x;   // < 2^32 (host check)
#pragma unroll
                for (int i = 0; i < TH; ++i) {
                    if (y0 + i < H) {
                        OutT *orow = orow0 + (size_t)i * W;   // uniform
                        orow[lane_off] = (OutT)CV::last(v, taps.lo, i);
                        orow[lane_off + band] = (OutT)CV::last(v, taps.hi, i);
                    }
                }
            }
            __syncthreads();
            slot_new += TH; slot_new = slot_new >= RH ? slot_new - RH : slot_new;
            s0 += TH; s0 = s0 >= RH ? s0 - RH : s0;
        }
    }
}

template <int L, int NLEV, int R, int TH, int NT, typename InT, int LAYOUT, bool BF16>
static int launch_slide(const void *in, void *out, SlideGeom g, const float *lo, const float *hi, hipStream_t st)
{
    constexpr int HALO = (L - 1) * ((1 << NLEV) - 1);
    constexpr int RH = TH + HALO;
    STaps<L> taps;
    for (int i = 0; i < L; ++i) { taps.lo[i] = lo[i]; taps.hi[i] = hi[i]; }
    g.nrun = (int)ceil_div(g.W, R);
    g.P = g.nrun * R + 4;                         // multiple of 4; rows hold whole runs
    const size_t lds = (size_t)2 * RH * g.P * sizeof(float);
    // shapes this kernel does not take: the caller falls back to the tiled kernels
    if (lds > (size_t)kMaxLdsBytes - 2048 || TH * g.nrun > NT) return 1;
    if (g.W < R + HALO || g.H < TH + HALO || (uint64_t)g.H * g.W * 4 >= (1ull << 30)) return 1;
    auto kern = k_swt_slide<L, NLEV, R, TH, NT, InT, LAYOUT, BF16>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "swt slide: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            WV_FAIL(WV_EHIP, "swt slide: device query failed");
        num_cu = prop.multiProcessorCount;
    }
    const int by_lds = std::max<int>(1, (int)((size_t)kMaxLdsBytes / (lds + 512)));
    const char *env = getenv("WV_SWT_WG_PER_CU");
    const int per_cu = env && atoi(env) > 0 ? atoi(env) : std::min(by_lds, 2048 / NT);
    const int64_t planes = (int64_t)g.B * g.C;
    const int64_t grid = std::min<int64_t>(planes, (int64_t)num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, (const InT *)in, out, g, taps);
    WV_CHECK_LAUNCH("k_swt_slide");
    return WV_OK;
}

template <int L, int NLEV, int R, int TH, int NT>
static int slide_types(const void *in, int in_dtype, void *out, const SlideGeom &g, const float *lo, const float *hi,
                       hipStream_t st)
{
    const bool nhwc3 = g.in_layout == WV_LAYOUT_NHWC && g.C == 3;
    if (g.in_layout != WV_LAYOUT_NCHW && !nhwc3) return 1;
#define WV_GO(T, LAY, BF) return launch_slide<L, NLEV, R, TH, NT, T, LAY, BF>(in, out, g, lo, hi, st)
    if (in_dtype == WV_DT_U8) {
        if (nhwc3) { if (g.out_bf16) WV_GO(uint8_t, 1, true); WV_GO(uint8_t, 1, false); }
        if (g.out_bf16) WV_GO(uint8_t, 0, true);
        WV_GO(uint8_t, 0, false);
    }
    if (nhwc3) { if (g.out_bf16) WV_GO(float, 1, true); WV_GO(float, 1, false); }
    if (g.out_bf16) WV_GO(float, 0, true);
    WV_GO(float, 0, false);
#undef WV_GO
}

bool swt_slide_covers(int L, int n, int W, int H)
{
    return ((L == 4 && n == 3) || (L == 2 && n == 1)) && (W % 4) == 0 && W <= 256 && W >= 40 && H >= 40;
}

int swt_slide_launch(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B, int C, int H,
                     int W, int n, const float *lo, const float *hi, int L, hipStream_t st)
{
    SlideGeom g{};
    g.B = B; g.C = C; g.H = H; g.W = W; g.in_layout = in_layout; g.out_bf16 = out_dtype == WV_DT_BF16;
    if (L == 4 && n == 3) return slide_types<4, 3, 16, 16, 256>(in, in_dtype, out, g, lo, hi, st);
    if (L == 2 && n == 1) return slide_types<2, 1, 16, 16, 256>(in, in_dtype, out, g, lo, hi, st);
    return 1;
}

}  // namespace wv

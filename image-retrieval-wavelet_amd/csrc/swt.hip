// Batched 2-D stationary wavelet transform for gfx950.
//
// What it computes (reference: pywt.swt2(...)[0] per channel at
// /root/reference/main/transforms/custom_transforms.py:163-166, wrapper :145-157):
//   for level l = 1..n, s = 2^(l-1), on A_{l-1} (A_0 = image/255):
//     1-D rule  y[o] = sum_m f[m] * x[(o + s*(L/2 - m)) mod N]   (periodized a-trous)
//     axis 0 (rows index) first, then axis 1;  aa=cA, da=cH, ad=cV, dd=cD
//   only the level-n bands are emitted, as out[b][c][{cA,cH,cV,cD}][H][W].
//
// Tiled kernel (k_swt_tiled): one workgroup owns a TH x TW tile of the output of one
// (image, channel) plane.  It loads the tile plus the halo every level needs
// ((L/2-1)(2^n-1) before, (L/2)(2^n-1) after, wrapped mod H / mod W at load time) into LDS
// as fp32, then runs all n levels inside LDS (ping-pong X <-> Y) and streams the four
// level-n bands out with 16-byte stores.  Intermediate levels never touch HBM, and the
// n-1 discarded detail levels are never computed.  HBM traffic = input once (+halo re-reads
// that hit L2) + output once: the algorithmic minimum of SURVEY.md 8(d).
//   vertical pass : thread = 4 adjacent columns (one ds_read_b128 per row) x R=4 outputs of one
//                   dilation phase, so R+L-1 LDS reads feed R*L MACs per column
//   horizontal pass: thread = 4 adjacent outputs of one row; the taps come from NV aligned
//                   float4 reads of that row (NV = 3..5) instead of 4*L scalar reads
// Generic kernels (k_axis_generic): any tap count / any size, one level per launch through a
// caller workspace; used only when the tiled kernel's constraints do not hold.
#include "common.hpp"
#include "swt_fused.hpp"

namespace wv {

constexpr int kMaxLevels = 4;
constexpr int kSwtThreads = 256;
constexpr int kVR = 4;  // vertical outputs per thread (same dilation phase)

template <int L>
struct Taps {
    float lo[L];
    float hi[L];
};

struct PassDesc {
    int rstart, nrows;        // output rows of this level (LDS row coordinates)
    int cg0, ngroups;         // output column groups (of 4) of the horizontal pass
    uint32_t ngroups_magic;   // ceil(2^32 / ngroups)
    int vblocks;              // ceil(ceil(nrows / s) / kVR)
};

struct SwtGeom {
    int B, C, H, W;
    int TH, TW, tilesX;
    int RH, RW;               // LDS region rows, row pitch in floats (multiple of 4)
    int HB, CB;               // halo rows before the tile, aligned column origin of the tile
    int G;                    // guard floats on both ends of every LDS buffer
    uint32_t ncg, ncg_magic;  // RW / 4
    int in_layout;
    PassDesc pass[kMaxLevels];
};

__device__ __forceinline__ uint32_t fast_div(uint32_t u, uint32_t d, uint32_t magic)
{
    return d == 1 ? u : __umulhi(u, magic);
}

__device__ __forceinline__ int wrap(int v, int n)
{
    while (v < 0) v += n;
    while (v >= n) v -= n;
    return v;
}

__device__ __forceinline__ float4 load4_as_f32(const uint8_t *p, int stride)
{
    float4 r;
    if (stride == 1) {
        uchar4 v = *reinterpret_cast<const uchar4 *>(p);
        r = make_float4((float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f,
                        (float)v.w / 255.0f);
    } else {
        r = make_float4((float)p[0] / 255.0f, (float)p[stride] / 255.0f,
                        (float)p[2 * stride] / 255.0f, (float)p[3 * stride] / 255.0f);
    }
    return r;
}

__device__ __forceinline__ float4 load4_as_f32(const float *p, int stride)
{
    if (stride == 1) return *reinterpret_cast<const float4 *>(p);
    return make_float4(p[0], p[stride], p[2 * stride], p[3 * stride]);
}

__device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__device__ __forceinline__ void store4(__hip_bfloat16 *p, float4 v)
{
    // plain casts: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved)
    union {
        __hip_bfloat16 h[4];
        uint2 u;
    } t;
    t.h[0] = __float2bfloat16(v.x);
    t.h[1] = __float2bfloat16(v.y);
    t.h[2] = __float2bfloat16(v.z);
    t.h[3] = __float2bfloat16(v.w);
    *reinterpret_cast<uint2 *>(p) = t.u;
}

__device__ __forceinline__ void f4_fma(float4 &acc, float coef, const float4 &val)
{
    acc.x = fmaf(coef, val.x, acc.x);
    acc.y = fmaf(coef, val.y, acc.y);
    acc.z = fmaf(coef, val.z, acc.z);
    acc.w = fmaf(coef, val.w, acc.w);
}
__device__ __forceinline__ float4 f4_mul(float coef, const float4 &val)
{
    return make_float4(coef * val.x, coef * val.y, coef * val.z, coef * val.w);
}

// ---------------------------------------------------------------------------- vertical pass
// dst_lo / dst_hi rows are indexed (row - lo_off) / (row - hi_off).
template <int L, int S, bool BOTH>
__device__ __forceinline__ void vpass(const float *__restrict__ src, float *__restrict__ dst_lo,
                                      float *__restrict__ dst_hi, int hi_off, const SwtGeom &g,
                                      const PassDesc &pd, const Taps<L> &taps)
{
    constexpr int NIN = kVR + L - 1;
    const uint32_t units = (uint32_t)S * pd.vblocks * g.ncg;
    for (uint32_t u = threadIdx.x; u < units; u += kSwtThreads) {
        const uint32_t t = fast_div(u, g.ncg, g.ncg_magic);
        const int cg = u - t * g.ncg;
        const int phi = t & (S - 1);
        const int blk = t / S;
        const int r0 = pd.rstart + phi + blk * (kVR * S);
        float4 in[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            int row = r0 - S * (L / 2 - 1) + i * S;
            row = min(max(row, 0), g.RH - 1);
            in[i] = *reinterpret_cast<const float4 *>(src + row * g.RW + 4 * cg);
        }
        const int rend = pd.rstart + pd.nrows;
#pragma unroll
        for (int j = 0; j < kVR; ++j) {
            const int row = r0 + j * S;
            if (row < rend) {
                float4 a = f4_mul(taps.lo[0], in[j + L - 1]);
#pragma unroll
                for (int m = 1; m < L; ++m) f4_fma(a, taps.lo[m], in[j + L - 1 - m]);
                *reinterpret_cast<float4 *>(dst_lo + row * g.RW + 4 * cg) = a;
                if (BOTH) {
                    float4 d = f4_mul(taps.hi[0], in[j + L - 1]);
#pragma unroll
                    for (int m = 1; m < L; ++m) f4_fma(d, taps.hi[m], in[j + L - 1 - m]);
                    *reinterpret_cast<float4 *>(dst_hi + (row - hi_off) * g.RW + 4 * cg) = d;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------- horizontal pass
template <int L, int S>
struct HGeom {
    static constexpr int before = S * (L / 2 - 1);
    static constexpr int after = S * (L / 2);
    static constexpr int PB = (before + 3) / 4 * 4;  // floats read before the group
    static constexpr int PA = (after + 3) / 4 * 4;   // floats read after the group
    static constexpr int NV = (PB + 4 + PA) / 4;     // aligned float4 reads
};

template <int L, int S>
__device__ __forceinline__ void hload(const float *row_c0, float (&v)[4 * HGeom<L, S>::NV])
{
    using HG = HGeom<L, S>;
#pragma unroll
    for (int i = 0; i < HG::NV; ++i) {
        float4 t = *reinterpret_cast<const float4 *>(row_c0 - HG::PB + 4 * i);
        v[4 * i + 0] = t.x;
        v[4 * i + 1] = t.y;
        v[4 * i + 2] = t.z;
        v[4 * i + 3] = t.w;
    }
}

template <int L, int S>
__device__ __forceinline__ float4 hfilter(const float (&v)[4 * HGeom<L, S>::NV], const float (&f)[L])
{
    using HG = HGeom<L, S>;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = f[0] * v[HG::PB + j + S * (L / 2)];
#pragma unroll
        for (int m = 1; m < L; ++m) a = fmaf(f[m], v[HG::PB + j + S * (L / 2 - m)], a);
        o[j] = a;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}

template <int L, int S>
__device__ __forceinline__ void hpass_mid(const float *__restrict__ src, float *__restrict__ dst,
                                          const SwtGeom &g, const PassDesc &pd,
                                          const Taps<L> &taps)
{
    const uint32_t units = (uint32_t)pd.nrows * pd.ngroups;
    for (uint32_t u = threadIdx.x; u < units; u += kSwtThreads) {
        const uint32_t rr = fast_div(u, pd.ngroups, pd.ngroups_magic);
        const int gi = u - rr * pd.ngroups;
        const int row = pd.rstart + rr;
        const int c0 = 4 * (pd.cg0 + gi);
        float v[4 * HGeom<L, S>::NV];
        hload<L, S>(src + row * g.RW + c0, v);
        *reinterpret_cast<float4 *>(dst + row * g.RW + c0) = hfilter<L, S>(v, taps.lo);
    }
}

template <int L, int S, typename OutT>
__device__ __forceinline__ void hpass_last(const float *__restrict__ ylo,
                                           const float *__restrict__ yhi, int hi_off,
                                           OutT *__restrict__ out_plane, int y0, int x0,
                                           const SwtGeom &g, const PassDesc &pd,
                                           const Taps<L> &taps)
{
    const uint32_t units = (uint32_t)pd.nrows * pd.ngroups;
    const size_t band = (size_t)g.H * g.W;
    for (uint32_t u = threadIdx.x; u < units; u += kSwtThreads) {
        const uint32_t rr = fast_div(u, pd.ngroups, pd.ngroups_magic);
        const int gi = u - rr * pd.ngroups;
        const int row = pd.rstart + rr;
        const int c0 = 4 * (pd.cg0 + gi);
        const int gy = y0 + (int)rr;
        const int gx = x0 + 4 * gi;
        if (gy >= g.H || gx >= g.W) continue;
        float v[4 * HGeom<L, S>::NV];
        OutT *o = out_plane + (size_t)gy * g.W + gx;
        hload<L, S>(ylo + row * g.RW + c0, v);
        store4(o, hfilter<L, S>(v, taps.lo));             // aa = cA
        store4(o + 2 * band, hfilter<L, S>(v, taps.hi));  // ad = cV
        hload<L, S>(yhi + (row - hi_off) * g.RW + c0, v);
        store4(o + band, hfilter<L, S>(v, taps.lo));      // da = cH
        store4(o + 3 * band, hfilter<L, S>(v, taps.hi));  // dd = cD
    }
}

template <int L, int NLEV, int LEV, typename OutT>
struct LevelRunner {
    static __device__ __forceinline__ void run(float *X, float *Y0, float *Y1, OutT *out_plane,
                                               int y0, int x0, const SwtGeom &g,
                                               const Taps<L> &taps)
    {
        constexpr int S = 1 << (LEV - 1);
        const PassDesc &pd = g.pass[LEV - 1];
        if constexpr (LEV < NLEV) {
            vpass<L, S, false>(X, Y0, nullptr, 0, g, pd, taps);
            __syncthreads();
            hpass_mid<L, S>(Y0, X, g, pd, taps);
            __syncthreads();
            LevelRunner<L, NLEV, LEV + 1, OutT>::run(X, Y0, Y1, out_plane, y0, x0, g, taps);
        } else {
            vpass<L, S, true>(X, Y0, Y1, g.HB, g, pd, taps);
            __syncthreads();
            hpass_last<L, S, OutT>(Y0, Y1, g.HB, out_plane, y0, x0, g, pd, taps);
        }
    }
};

template <int L, int NLEV, typename InT, typename OutT>
__global__ __launch_bounds__(kSwtThreads) void k_swt_tiled(const InT *__restrict__ in,
                                                           OutT *__restrict__ out, SwtGeom g,
                                                           Taps<L> taps)
{
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    const int bufsz = g.RH * g.RW + 2 * g.G;
    float *X = lds + g.G;
    float *Y0 = X + bufsz;
    float *Y1 = Y0 + bufsz;

    const int tile = blockIdx.x;
    const int ty = tile / g.tilesX;
    const int tx = tile - ty * g.tilesX;
    const int c = blockIdx.y, b = blockIdx.z;
    const int x0 = tx * g.TW, y0 = ty * g.TH;

    // ---- global -> LDS, wrapped, converted to fp32 in [0,1]
    {
        const uint32_t units = (uint32_t)g.RH * g.ncg;
        const int gx_base = x0 - g.CB, gy_base = y0 - g.HB;
        for (uint32_t u = threadIdx.x; u < units; u += kSwtThreads) {
            const uint32_t r = fast_div(u, g.ncg, g.ncg_magic);
            const int cg = u - r * g.ncg;
            const int gy = wrap(gy_base + (int)r, g.H);
            const int gx = wrap(gx_base + 4 * cg, g.W);
            float4 v;
            if (g.in_layout == WV_LAYOUT_NCHW) {
                const InT *p = in + (((size_t)b * g.C + c) * g.H + gy) * g.W + gx;
                v = load4_as_f32(p, 1);
            } else {
                const InT *p = in + (((size_t)b * g.H + gy) * g.W + gx) * g.C + c;
                v = load4_as_f32(p, g.C);
            }
            *reinterpret_cast<float4 *>(X + r * g.RW + 4 * cg) = v;
        }
    }
    __syncthreads();
    OutT *out_plane = out + ((size_t)b * g.C + c) * 4 * (size_t)g.H * g.W;
    LevelRunner<L, NLEV, 1, OutT>::run(X, Y0, Y1, out_plane, y0, x0, g, taps);
}

// ----------------------------------------------------------------------------- generic path
struct GenTaps {
    float lo[32];
    float hi[32];
};

template <typename InT>
__global__ void k_planes_to_f32(const InT *__restrict__ in, float *__restrict__ dst, int B, int C,
                                int H, int W, int in_layout)
{
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t x = i % W, y = (i / W) % H, c = (i / ((size_t)W * H)) % C, b = i / ((size_t)W * H * C);
        size_t src = in_layout == WV_LAYOUT_NCHW ? i : ((b * H + y) * W + x) * C + c;
        float v;
        if constexpr (sizeof(InT) == 1) v = (float)in[src] / 255.0f;
        else v = (float)in[src];
        dst[i] = v;
    }
}

// One periodized a-trous pass along `axis` of P planes [H][W].  dst_* strides let the last level
// write straight into out[b][c][band].
template <typename OutT>
__global__ void k_axis_generic(const float *__restrict__ src, OutT *__restrict__ dst_lo,
                               OutT *__restrict__ dst_hi, size_t dst_plane_stride, int P, int H,
                               int W, int axis, int s, int L, GenTaps taps)
{
    const size_t hw = (size_t)H * W;
    const size_t total = (size_t)P * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = i % W, y = (i / W) % H;
        const size_t p = i / hw;
        const float *pl = src + p * hw;
        const int n = axis == 0 ? H : W;
        const int o = axis == 0 ? y : x;
        float a = 0.f, d = 0.f;
        for (int m = 0; m < L; ++m) {
            int idx = (int)(((long)o + (long)s * (L / 2 - m)) % n);
            if (idx < 0) idx += n;
            const float v = axis == 0 ? pl[(size_t)idx * W + x] : pl[(size_t)y * W + idx];
            a = m == 0 ? taps.lo[0] * v : fmaf(taps.lo[m], v, a);
            d = m == 0 ? taps.hi[0] * v : fmaf(taps.hi[m], v, d);
        }
        const size_t o_off = p * dst_plane_stride + (size_t)y * W + x;
        if (dst_lo) dst_lo[o_off] = (OutT)a;
        if (dst_hi) dst_hi[o_off] = (OutT)d;
    }
}

template <typename InT, typename OutT>
__global__ void k_rawstack(const InT *__restrict__ in, OutT *__restrict__ out, int B, int C, int H,
                           int W, int copies, int in_layout)
{
    const size_t hw = (size_t)H * W;
    const size_t total = (size_t)B * C * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t x = i % W, y = (i / W) % H, c = (i / hw) % C, b = i / (hw * C);
        size_t src = in_layout == WV_LAYOUT_NCHW ? i : ((b * H + y) * W + x) * C + c;
        float v;
        if constexpr (sizeof(InT) == 1) v = (float)in[src] / 255.0f;
        else v = (float)in[src];
        OutT *o = out + ((b * C + c) * copies) * hw + y * W + x;
        for (int k = 0; k < copies; ++k) o[k * hw] = (OutT)v;
    }
}

// ------------------------------------------------------------------------------- host side
static uint32_t magic_of(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }

struct TilePlan {
    bool ok;
    SwtGeom g;
    size_t lds_bytes;
};

static size_t tile_lds_bytes(int L, int n, int TH, int TW, SwtGeom *out)
{
    const int span = (1 << n) - 1;
    const int HB = (L / 2 - 1) * span, HA = (L / 2) * span;
    const int CB = (int)align_up(HB, 4);
    const int RW = (int)align_up(CB + TW + HA, 4);
    const int RH = HB + TH + HA;
    const int G = (int)align_up((1 << (n - 1)) * (L / 2) + 8, 4);
    if (out) {
        out->TH = TH; out->TW = TW; out->HB = HB; out->CB = CB; out->RW = RW; out->RH = RH; out->G = G;
    }
    return ((size_t)2 * ((size_t)RH * RW + 2 * G) + (size_t)TH * RW + 2 * G) * sizeof(float);
}

static TilePlan plan_tiles(int B, int C, int H, int W, int L, int n, int in_layout)
{
    TilePlan p{};
    p.ok = false;
    if (!(L == 2 || L == 4 || L == 8 || L == 10) || n < 1 || n > 3 || (W % 4) != 0) return p;
    // column tiles: split W evenly into pieces of <= 128 (multiples of 4)
    int tilesX = (int)ceil_div(W, 128);
    int TW = (int)align_up(ceil_div(W, tilesX), 4);
    int TH = 32;
    const char *env = ::wv::tune("WV_SWT_TILE");  // "TH,TW" override for tuning
    int eth = 0, etw = 0;
    if (env && sscanf(env, "%d,%d", &eth, &etw) == 2 && eth > 0 && etw > 0 && etw % 4 == 0) {
        TH = eth; TW = etw;
    } else {
        if (TH > H) TH = H;
        const size_t two_per_cu = 78 * 1024;
        while (TH > 8 && tile_lds_bytes(L, n, TH, TW, nullptr) > two_per_cu) TH -= 8;
        if (tile_lds_bytes(L, n, TH, TW, nullptr) > two_per_cu) {
            TH = H < 32 ? H : 32;  // accept one workgroup per CU
            while (TH > 4 && tile_lds_bytes(L, n, TH, TW, nullptr) > (size_t)kMaxLdsBytes - 1024) TH -= 4;
        }
    }
    if (TW > W) TW = W;
    tilesX = (int)ceil_div(W, TW);
    SwtGeom &g = p.g;
    p.lds_bytes = tile_lds_bytes(L, n, TH, TW, &g);
    if (p.lds_bytes > (size_t)kMaxLdsBytes - 1024) return p;
    g.B = B; g.C = C; g.H = H; g.W = W; g.tilesX = tilesX; g.in_layout = in_layout;
    g.ncg = g.RW / 4; g.ncg_magic = magic_of(g.ncg);
    if ((uint64_t)g.RH * g.ncg * 4 >= (1ull << 32) / (g.ncg + 1)) return p;  // fast_div range
    for (int l = 1; l <= n; ++l) {
        PassDesc &pd = g.pass[l - 1];
        const int rem = (1 << n) - (1 << l);
        const int hb = (L / 2 - 1) * rem, ha = (L / 2) * rem;
        pd.rstart = g.HB - hb;
        pd.nrows = TH + hb + ha;
        const int c_lo = (g.CB - hb) / 4 * 4;
        const int c_hi = (int)align_up(g.CB + TW + ha, 4);
        pd.cg0 = c_lo / 4;
        pd.ngroups = (c_hi - c_lo) / 4;
        pd.ngroups_magic = magic_of(pd.ngroups);
        const int s = 1 << (l - 1);
        pd.vblocks = (int)ceil_div(ceil_div(pd.nrows, s), kVR);
    }
    p.ok = true;
    return p;
}

template <int L, int NLEV, typename InT, typename OutT>
static int launch_tiled(const void *in, void *out, const TilePlan &p, const float *lo,
                        const float *hi, hipStream_t st)
{
    Taps<L> taps;
    for (int i = 0; i < L; ++i) { taps.lo[i] = lo[i]; taps.hi[i] = hi[i]; }
    auto kern = k_swt_tiled<L, NLEV, InT, OutT>;
    if (p.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)p.lds_bytes);
        if (e != hipSuccess) WV_FAIL(WV_EHIP, "swt: hipFuncSetAttribute(%zu B LDS): %s", p.lds_bytes,
                                     hipGetErrorString(e));
    }
    const SwtGeom &g = p.g;
    dim3 grid((unsigned)(g.tilesX * ceil_div(g.H, g.TH)), (unsigned)g.C, (unsigned)g.B);
    hipLaunchKernelGGL(kern, grid, dim3(kSwtThreads), p.lds_bytes, st, (const InT *)in, (OutT *)out,
                       g, taps);
    WV_CHECK_LAUNCH("k_swt_tiled");
    return WV_OK;
}

template <int L, typename InT, typename OutT>
static int dispatch_levels(int n, const void *in, void *out, const TilePlan &p, const float *lo,
                           const float *hi, hipStream_t st)
{
    switch (n) {
    case 1: return launch_tiled<L, 1, InT, OutT>(in, out, p, lo, hi, st);
    case 2: return launch_tiled<L, 2, InT, OutT>(in, out, p, lo, hi, st);
    case 3: return launch_tiled<L, 3, InT, OutT>(in, out, p, lo, hi, st);
    }
    WV_FAIL(WV_ENOTSUP, "swt tiled: level %d", n);
}

template <typename InT, typename OutT>
static int dispatch_taps(int L, int n, const void *in, void *out, const TilePlan &p, const float *lo,
                         const float *hi, hipStream_t st)
{
    switch (L) {
    case 2: return dispatch_levels<2, InT, OutT>(n, in, out, p, lo, hi, st);
    case 4: return dispatch_levels<4, InT, OutT>(n, in, out, p, lo, hi, st);
    case 8: return dispatch_levels<8, InT, OutT>(n, in, out, p, lo, hi, st);
    case 10: return dispatch_levels<10, InT, OutT>(n, in, out, p, lo, hi, st);
    }
    WV_FAIL(WV_ENOTSUP, "swt tiled: %d taps", L);
}

static int grid_for(size_t total) { return (int)std::min<size_t>(ceil_div((int64_t)total, 256), 256 * 16); }

template <typename InT, typename OutT>
static int run_generic(const void *in, void *out, int B, int C, int H, int W, int n, const float *lo,
                       const float *hi, int L, int in_layout, void *ws, size_t ws_bytes,
                       hipStream_t st)
{
    const size_t P = (size_t)B * C, hw = (size_t)H * W, plane_elems = P * hw;
    if (ws_bytes < 3 * plane_elems * sizeof(float) || !ws)
        WV_FAIL(WV_ENOMEM, "swt generic path needs %zu workspace bytes, got %zu",
                3 * plane_elems * sizeof(float), ws_bytes);
    if (L > 32) WV_FAIL(WV_ENOTSUP, "swt: more than 32 taps (%d)", L);
    GenTaps taps{};
    for (int i = 0; i < L; ++i) { taps.lo[i] = lo[i]; taps.hi[i] = hi[i]; }
    float *cur = (float *)ws, *ta = cur + plane_elems, *td = ta + plane_elems;
    const int grid = grid_for(plane_elems);
    hipLaunchKernelGGL((k_planes_to_f32<InT>), dim3(grid), dim3(256), 0, st, (const InT *)in, cur, B,
                       C, H, W, in_layout);
    OutT *o = (OutT *)out;
    for (int l = 1; l <= n; ++l) {
        const int s = 1 << (l - 1);
        const bool last = l == n;
        hipLaunchKernelGGL((k_axis_generic<float>), dim3(grid), dim3(256), 0, st, cur, ta,
                           last ? td : (float *)nullptr, hw, (int)P, H, W, 0, s, L, taps);
        if (!last) {
            hipLaunchKernelGGL((k_axis_generic<float>), dim3(grid), dim3(256), 0, st, ta, cur,
                               (float *)nullptr, hw, (int)P, H, W, 1, s, L, taps);
        } else {
            hipLaunchKernelGGL((k_axis_generic<OutT>), dim3(grid), dim3(256), 0, st, ta, o, o + 2 * hw,
                               4 * hw, (int)P, H, W, 1, s, L, taps);
            hipLaunchKernelGGL((k_axis_generic<OutT>), dim3(grid), dim3(256), 0, st, td, o + hw,
                               o + 3 * hw, 4 * hw, (int)P, H, W, 1, s, L, taps);
        }
    }
    WV_CHECK_LAUNCH("swt generic");
    return WV_OK;
}

template <typename InT, typename OutT>
static int swt_typed(const void *in, void *out, int B, int C, int H, int W, int n, const float *lo,
                     const float *hi, int L, int in_layout, void *ws, size_t ws_bytes, hipStream_t st)
{
    // WV_SWT_PATH = fused | tiled | generic pins one implementation (tests / tuning); default: best available
    const char *path = ::wv::tune("WV_SWT_PATH");
    const bool want_slide = !path || !strcmp(path, "slide");
    const bool want_fused = !path || !strcmp(path, "fused");
    const bool want_tiled = !path || !strcmp(path, "tiled");
    if (want_slide && swt_slide_covers(L, n, W, H)) {
        const int rc = swt_slide_launch(in, sizeof(InT) == 1 ? WV_DT_U8 : WV_DT_F32, in_layout, out,
                                        sizeof(OutT) == 2 ? WV_DT_BF16 : WV_DT_F32, B, C, H, W, n, lo, hi, L, st);
        if (rc <= 0) return rc;
    }
    if (want_fused && swt_fused_covers(L, n, W)) {
        const int rc = swt_fused_launch(in, sizeof(InT) == 1 ? WV_DT_U8 : WV_DT_F32, in_layout, out,
                                        sizeof(OutT) == 2 ? WV_DT_BF16 : WV_DT_F32, B, C, H, W, n, lo, hi, L, st);
        if (rc <= 0) return rc;
    }
    TilePlan p = plan_tiles(B, C, H, W, L, n, in_layout);
    if (want_tiled && p.ok) return dispatch_taps<InT, OutT>(L, n, in, out, p, lo, hi, st);
    return run_generic<InT, OutT>(in, out, B, C, H, W, n, lo, hi, L, in_layout, ws, ws_bytes, st);
}

}  // namespace wv

using namespace wv;

extern "C" size_t wv_swt2d_workspace_bytes(int B, int C, int H, int W, int level, int flen)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const char *path = ::wv::tune("WV_SWT_PATH");
    if (!path || strcmp(path, "generic")) {
        if (swt_slide_covers(flen, level, W, H) && (!path || !strcmp(path, "slide"))) return 0;
        if (swt_fused_covers(flen, level, W) && (!path || !strcmp(path, "fused"))) return 0;
        TilePlan p = plan_tiles(B, C, H, W, flen, level, WV_LAYOUT_NCHW);
        if (p.ok && (!path || !strcmp(path, "tiled"))) return 0;
    }
    return (size_t)3 * B * C * H * W * sizeof(float);
}

extern "C" int wv_swt2d_forward(const void *in, int in_dtype, int in_layout, void *out, int out_dtype,
                                int B, int C, int H, int W, int level, const float *dec_lo,
                                const float *dec_hi, int flen, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(in && out, "swt: null buffer");
    WV_REQUIRE(dec_lo && dec_hi && flen >= 1, "swt: missing filter taps");
    WV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "swt: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    WV_REQUIRE(level >= 1 && level <= 12, "swt: level %d out of range", level);
    WV_REQUIRE((H % (1 << level)) == 0 && (W % (1 << level)) == 0,
               "swt: H=%d, W=%d must be multiples of 2^level=%d (PyWavelets raises ValueError here)",
               H, W, 1 << level);
    WV_REQUIRE(in_layout == WV_LAYOUT_NCHW || in_layout == WV_LAYOUT_NHWC, "swt: bad layout %d",
               in_layout);
    WV_REQUIRE(B <= 65535 && C <= 65535, "swt: B and C must be <= 65535 per call");
    hipStream_t st = (hipStream_t)stream;
    if (in_dtype == WV_DT_U8 && out_dtype == WV_DT_F32)
        return swt_typed<uint8_t, float>(in, out, B, C, H, W, level, dec_lo, dec_hi, flen, in_layout,
                                         workspace, workspace_bytes, st);
    if (in_dtype == WV_DT_F32 && out_dtype == WV_DT_F32)
        return swt_typed<float, float>(in, out, B, C, H, W, level, dec_lo, dec_hi, flen, in_layout,
                                       workspace, workspace_bytes, st);
    if (in_dtype == WV_DT_U8 && out_dtype == WV_DT_BF16)
        return swt_typed<uint8_t, __hip_bfloat16>(in, out, B, C, H, W, level, dec_lo, dec_hi, flen,
                                                  in_layout, workspace, workspace_bytes, st);
    if (in_dtype == WV_DT_F32 && out_dtype == WV_DT_BF16)
        return swt_typed<float, __hip_bfloat16>(in, out, B, C, H, W, level, dec_lo, dec_hi, flen,
                                                in_layout, workspace, workspace_bytes, st);
    WV_FAIL(WV_ENOTSUP, "swt: dtype pair in=%d out=%d not supported", in_dtype, out_dtype);
}

extern "C" int wv_swt2d_forward_ex(const void *in, int in_dtype, int in_layout, void *out, int out_dtype,
                                   int out_layout, int64_t band_stride, int B, int C, int H, int W, int level,
                                   const float *dec_lo, const float *dec_hi, int flen, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (out_layout == WV_BANDS_INNER)
        return wv_swt2d_forward(in, in_dtype, in_layout, out, out_dtype, B, C, H, W, level, dec_lo, dec_hi, flen,
                                workspace, workspace_bytes, stream);
    WV_REQUIRE(out_layout == WV_BANDS_OUTER, "swt: bad output layout %d", out_layout);
    WV_REQUIRE(in && out, "swt: null buffer");
    WV_REQUIRE(dec_lo && dec_hi && flen >= 1, "swt: missing filter taps");
    WV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "swt: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    WV_REQUIRE(level >= 1 && level <= 12, "swt: level %d out of range", level);
    WV_REQUIRE((H % (1 << level)) == 0 && (W % (1 << level)) == 0,
               "swt: H=%d, W=%d must be multiples of 2^level=%d (PyWavelets raises ValueError here)",
               H, W, 1 << level);
    WV_REQUIRE(in_layout == WV_LAYOUT_NCHW || in_layout == WV_LAYOUT_NHWC, "swt: bad layout %d", in_layout);
    WV_REQUIRE(B <= 65535 && C <= 65535, "swt: B and C must be <= 65535 per call");
    WV_REQUIRE(band_stride >= (int64_t)B * C * H * W, "swt: band_stride %lld < B*C*H*W", (long long)band_stride);
    WV_REQUIRE((in_dtype == WV_DT_U8 || in_dtype == WV_DT_F32) && (out_dtype == WV_DT_F32 || out_dtype == WV_DT_BF16),
               "swt: dtype pair in=%d out=%d not supported", in_dtype, out_dtype);
    // band-major output exists in the sliding kernel only (the shapes of the hot path); the caller re-lays the
    // reference layout out for anything else
    if (!swt_slide_covers(flen, level, W, H))
        WV_FAIL(WV_ENOTSUP, "swt: band-major output is implemented by the sliding kernel only (W %% 4 == 0, 40 <= W <= 256, "
                            "H >= 40, 2/4 taps at levels 1-3 or 8/10 taps at level 1)");
    const int rc = swt_slide_launch(in, in_dtype, in_layout, out, out_dtype, B, C, H, W, level, dec_lo, dec_hi, flen,
                                    (hipStream_t)stream, WV_BANDS_OUTER, band_stride);
    if (rc > 0) WV_FAIL(WV_ENOTSUP, "swt: shape outside the sliding kernel's window");
    return rc;
}

extern "C" int wv_rawstack_forward(const void *in, int in_dtype, int in_layout, void *out,
                                   int out_dtype, int B, int C, int H, int W, int copies,
                                   void *stream)
{
    WV_REQUIRE(in && out, "rawstack: null buffer");
    WV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && copies > 0, "rawstack: bad shape");
    WV_REQUIRE(out_dtype == WV_DT_F32, "rawstack: only f32 output");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * C * H * W;
    const int grid = grid_for(total);
    if (in_dtype == WV_DT_U8)
        hipLaunchKernelGGL((k_rawstack<uint8_t, float>), dim3(grid), dim3(256), 0, st,
                           (const uint8_t *)in, (float *)out, B, C, H, W, copies, in_layout);
    else if (in_dtype == WV_DT_F32)
        hipLaunchKernelGGL((k_rawstack<float, float>), dim3(grid), dim3(256), 0, st, (const float *)in,
                           (float *)out, B, C, H, W, copies, in_layout);
    else
        WV_FAIL(WV_ENOTSUP, "rawstack: input dtype %d", in_dtype);
    WV_CHECK_LAUNCH("k_rawstack");
    return WV_OK;
}

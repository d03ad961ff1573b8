// Decimated multi-level 2-D DWT (DWTTransform) for gfx950 -- completeness of the transform plugin family,
// not a tuned kernel: one thread per output coefficient, one launch per level and axis, through a caller
// workspace.
//
// Reference: DWTTransform._apply_wavelet (/root/reference/main/transforms/custom_transforms.py:197-201):
//   coeffs = pywt.wavedec2(channel, wavelet, level);  cA = coeffs[0];  cH, cV, cD = coeffs[1]
// i.e. the approximation and the three detail bands of the COARSEST level.  PyWavelets' default signal
// extension is mode='symmetric' (half-sample: x[-1] = x[0], x[N] = x[N-1], ...), and its decimating convolution is
//   y[o] = sum_{j=0}^{F-1} f[j] * x_ext[2*o + 1 - j],   o = 0 .. floor((N + F - 1) / 2) - 1
// applied along axis 0 then axis 1 per level (keys 'aa','da','ad','dd' = cA,cH,cV,cD, first letter = axis 0).
#include "common.hpp"

namespace wv {

struct DwtTaps {
    float lo[32];
    float hi[32];
};

__host__ __device__ inline int dwt_len(int n, int flen) { return (n + flen - 1) / 2; }

__device__ __forceinline__ int sym_index(int i, int n)
{
    // half-sample symmetric extension, repeated as often as needed (signals shorter than the filter)
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

template <typename InT>
__global__ void k_dwt_planes_to_f32(const InT *__restrict__ in, float *__restrict__ dst, int B, int C, int H, int W,
                                    int in_layout)
{
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t x = i % W, y = (i / W) % H, c = (i / ((size_t)W * H)) % C, b = i / ((size_t)W * H * C);
        size_t src = in_layout == WV_LAYOUT_NCHW ? i : ((b * H + y) * W + x) * C + c;
        float v;
        if constexpr (sizeof(InT) == 1) v = (float)in[src] / 255.0f;
        else v = (float)in[src];
        dst[i] = v;
    }
}

// src [P][h][w] -> dst_lo / dst_hi [P][ho][wo] where the filtered axis is halved (with extension growth)
__global__ void k_dwt_axis(const float *__restrict__ src, float *__restrict__ dst_lo, float *__restrict__ dst_hi,
                           size_t dst_plane_stride, int P, int h, int w, int axis, int L, DwtTaps taps)
{
    const int ho = axis == 0 ? dwt_len(h, L) : h, wo = axis == 1 ? dwt_len(w, L) : w;
    const size_t total = (size_t)P * ho * wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = i % wo, y = (i / wo) % ho;
        const size_t p = i / ((size_t)wo * ho);
        const float *pl = src + p * (size_t)h * w;
        const int n = axis == 0 ? h : w, o = axis == 0 ? y : x;
        float a = 0.f, d = 0.f;
        for (int j = 0; j < L; ++j) {
            const int idx = sym_index(2 * o + 1 - j, n);
            const float v = axis == 0 ? pl[(size_t)idx * w + x] : pl[(size_t)y * w + idx];
            a = j == 0 ? taps.lo[0] * v : fmaf(taps.lo[j], v, a);
            d = j == 0 ? taps.hi[0] * v : fmaf(taps.hi[j], v, d);
        }
        const size_t off = p * dst_plane_stride + (size_t)y * wo + x;
        if (dst_lo) dst_lo[off] = a;
        if (dst_hi) dst_hi[off] = d;
    }
}

static int dwt_grid(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 256 * 16); }

}  // namespace wv

using namespace wv;

extern "C" int wv_dwt_out_len(int n, int flen, int level)
{
    for (int l = 0; l < level; ++l) n = dwt_len(n, flen);
    return n;
}

extern "C" size_t wv_dwt2d_workspace_bytes(int B, int C, int H, int W, int level, int flen)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || level < 1) return 0;
    // level-0 image + (lo, hi) after the axis-0 pass of level 1 + the next approximation: all bounded by H*W
    const size_t plane = (size_t)(H + flen) * (W + flen);
    return (size_t)B * C * plane * 4 * sizeof(float);
}

extern "C" int wv_dwt2d_forward(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W,
                                int level, const float *dec_lo, const float *dec_hi, int flen, void *workspace,
                                size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(in && out && dec_lo && dec_hi, "dwt: null buffer");
    WV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "dwt: bad shape");
    WV_REQUIRE(level >= 1 && level <= 12, "dwt: level %d out of range", level);
    WV_REQUIRE(flen >= 2 && flen <= 32, "dwt: %d taps (supported: 2..32)", flen);
    WV_REQUIRE(in_dtype == WV_DT_U8 || in_dtype == WV_DT_F32, "dwt: input dtype %d", in_dtype);
    const size_t need = wv_dwt2d_workspace_bytes(B, C, H, W, level, flen);
    if (!workspace || workspace_bytes < need) WV_FAIL(WV_ENOMEM, "dwt: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    DwtTaps taps{};
    for (int i = 0; i < flen; ++i) { taps.lo[i] = dec_lo[i]; taps.hi[i] = dec_hi[i]; }
    const size_t P = (size_t)B * C, slot = (size_t)(H + flen) * (W + flen) * P;
    float *cur = (float *)workspace, *ta = cur + slot, *td = ta + slot, *nxt = td + slot;
    const size_t n0 = P * H * W;
    if (in_dtype == WV_DT_U8)
        hipLaunchKernelGGL((k_dwt_planes_to_f32<uint8_t>), dim3(dwt_grid(n0)), dim3(256), 0, st, (const uint8_t *)in, cur, B, C, H, W, in_layout);
    else
        hipLaunchKernelGGL((k_dwt_planes_to_f32<float>), dim3(dwt_grid(n0)), dim3(256), 0, st, (const float *)in, cur, B, C, H, W, in_layout);
    int h = H, w = W;
    for (int l = 1; l <= level; ++l) {
        const int h2 = dwt_len(h, flen), w2 = dwt_len(w, flen);
        const bool last = l == level;
        // axis 0: cur [P][h][w] -> ta, td [P][h2][w]
        hipLaunchKernelGGL(k_dwt_axis, dim3(dwt_grid(P * h2 * w)), dim3(256), 0, st, cur, ta, last ? td : (float *)nullptr,
                           (size_t)h2 * w, (int)P, h, w, 0, flen, taps);
        if (!last) {   // only the approximation feeds the next level
            hipLaunchKernelGGL(k_dwt_axis, dim3(dwt_grid(P * h2 * w2)), dim3(256), 0, st, ta, nxt, (float *)nullptr,
                               (size_t)h2 * w2, (int)P, h2, w, 1, flen, taps);
            std::swap(cur, nxt);
        } else {       // write the four coarsest bands straight into out[b][c][band]
            const size_t band = (size_t)h2 * w2;
            hipLaunchKernelGGL(k_dwt_axis, dim3(dwt_grid(P * band)), dim3(256), 0, st, ta, out, out + 2 * band, 4 * band,
                               (int)P, h2, w, 1, flen, taps);            // aa = cA, ad = cV
            hipLaunchKernelGGL(k_dwt_axis, dim3(dwt_grid(P * band)), dim3(256), 0, st, td, out + band, out + 3 * band,
                               4 * band, (int)P, h2, w, 1, flen, taps);  // da = cH, dd = cD
        }
        h = h2; w = w2;
    }
    WV_CHECK_LAUNCH("dwt");
    return WV_OK;
}

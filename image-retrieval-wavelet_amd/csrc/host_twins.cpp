// Host twins of the transform entry points: the same arithmetic as the gfx950 kernels on HOST pointers.
//
// Why they exist ("who calls it / threading"): the reference runs its wavelet transform inside
// forked DataLoader worker processes, one PIL image at a time (/root/reference/main/datasets/flikr_coco.py:59-60
// -> custom_transforms.py:145-157), and a forked worker cannot use the parent's GPU context.  With an UNCHANGED
// transform YAML and an unchanged DataLoader(num_workers > 0) the plugin's __call__ therefore lands here.
// Contract: no HIP call, no thread is started, no global state -> safe after fork(); plain C++ compiled for the host.
//
// Arithmetic: float32, the order of operations of the device kernels (swt_slide.hip / swt_fused.hip): all levels of
// the row direction (axis 1) first, then all levels of the column direction (axis 0); per 1-D pass
//     y[o] = sum_m f[m] * x[(o + 2^(l-1) * (L/2 - m)) mod N],  a = f[0]*x0, then a = fma(f[m], x_m, a), m = 1..L-1
// so that the two implementations agree bit for bit on the shapes the sliding kernel covers.  (The separable passes
// of the two axes commute; PyWavelets interleaves them per level -- the difference is fp32 rounding, see DESIGN.md.)
// This file is product code; it shares nothing with oracle/.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/wvhash.h"

namespace wv {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
}

#define HT_FAIL(code, ...)            \
    do {                              \
        ::wv::set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)
#define HT_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) HT_FAIL(WV_EINVAL, __VA_ARGS__); \
    } while (0)

namespace {

#define HT_INLINE static inline __attribute__((always_inline))

// FMA = true: compiled inside a target("avx2,fma") caller -> vfmadd; false: libm fmaf (exact as well, slower).
template <bool FMA>
HT_INLINE float fma_(float a, float b, float c)
{
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    else return fmaf(a, b, c);
}

struct Taps {
    int L;
    float lo[32], hi[32];
};

// One periodized a-trous pass along x over a whole plane: src [H][W] -> dlo (and dhi when non-null).
template <bool FMA>
HT_INLINE void pass_rows(const float *src, float *dlo, float *dhi, int H, int W, const Taps &t, int S, float *pad)
{
    const int L = t.L;
    int off[32];
    int dmin = 0, dmax = 0;
    for (int m = 0; m < L; ++m) {
        off[m] = S * (L / 2 - m);
        dmin = off[m] < dmin ? off[m] : dmin;
        dmax = off[m] > dmax ? off[m] : dmax;
    }
    const int left = -dmin, right = dmax;
    for (int y = 0; y < H; ++y) {
        const float *row = src + (size_t)y * W;
        for (int i = 0; i < left; ++i) pad[i] = row[((i - left) % W + W) % W];
        memcpy(pad + left, row, (size_t)W * sizeof(float));
        for (int i = 0; i < right; ++i) pad[left + W + i] = row[i % W];
        const float *p = pad + left;
        float *ol = dlo + (size_t)y * W;
        {
            const float *x0 = p + off[0];
            const float f0 = t.lo[0];
            for (int o = 0; o < W; ++o) ol[o] = f0 * x0[o];
            for (int m = 1; m < L; ++m) {
                const float *xm = p + off[m];
                const float fm = t.lo[m];
                for (int o = 0; o < W; ++o) ol[o] = fma_<FMA>(fm, xm[o], ol[o]);
            }
        }
        if (dhi) {
            float *oh = dhi + (size_t)y * W;
            const float *x0 = p + off[0];
            const float f0 = t.hi[0];
            for (int o = 0; o < W; ++o) oh[o] = f0 * x0[o];
            for (int m = 1; m < L; ++m) {
                const float *xm = p + off[m];
                const float fm = t.hi[m];
                for (int o = 0; o < W; ++o) oh[o] = fma_<FMA>(fm, xm[o], oh[o]);
            }
        }
    }
}

// The same pass along y: output row y reads input rows (y + off[m]) mod H.  dst rows may have their own pitch
// (the last level writes straight into the output bands).
template <bool FMA>
HT_INLINE void pass_cols(const float *src, float *dlo, float *dhi, int H, int W, const Taps &t, int S)
{
    const int L = t.L;
    for (int y = 0; y < H; ++y) {
        const float *rows[32];
        for (int m = 0; m < L; ++m) {
            int r = (y + S * (L / 2 - m)) % H;
            rows[m] = src + (size_t)(r < 0 ? r + H : r) * W;
        }
        float *ol = dlo + (size_t)y * W;
        {
            const float f0 = t.lo[0];
            const float *x0 = rows[0];
            for (int x = 0; x < W; ++x) ol[x] = f0 * x0[x];
            for (int m = 1; m < L; ++m) {
                const float fm = t.lo[m];
                const float *xm = rows[m];
                for (int x = 0; x < W; ++x) ol[x] = fma_<FMA>(fm, xm[x], ol[x]);
            }
        }
        if (dhi) {
            float *oh = dhi + (size_t)y * W;
            const float f0 = t.hi[0];
            const float *x0 = rows[0];
            for (int x = 0; x < W; ++x) oh[x] = f0 * x0[x];
            for (int m = 1; m < L; ++m) {
                const float fm = t.hi[m];
                const float *xm = rows[m];
                for (int x = 0; x < W; ++x) oh[x] = fma_<FMA>(fm, xm[x], oh[x]);
            }
        }
    }
}

struct Scratch {
    std::vector<float> a, b, c, d, pad;
};

// plane x [H][W] (already float) -> out [4][H][W] = cA, cH, cV, cD of level n
template <bool FMA>
HT_INLINE void swt_plane(const float *x, float *out, int H, int W, int n, const Taps &t, Scratch &s)
{
    const size_t hw = (size_t)H * W;
    float *cur = s.a.data(), *nxt = s.b.data(), *rlo = s.c.data(), *rhi = s.d.data();
    const float *src = x;
    // row direction: levels 1..n-1 keep only the low-pass chain, level n yields (row-lo, row-hi)
    for (int l = 1; l < n; ++l) {
        pass_rows<FMA>(src, cur, nullptr, H, W, t, 1 << (l - 1), s.pad.data());
        src = cur;
        float *tmp = cur; cur = nxt; nxt = tmp;
    }
    pass_rows<FMA>(src, rlo, rhi, H, W, t, 1 << (n - 1), s.pad.data());
    // column direction on both row-filtered planes
    float *planes[2] = {rlo, rhi};
    for (int pl = 0; pl < 2; ++pl) {
        const float *p = planes[pl];
        float *u = s.a.data(), *v = s.b.data();
        for (int l = 1; l < n; ++l) {
            pass_cols<FMA>(p, u, nullptr, H, W, t, 1 << (l - 1));
            p = u;
            float *tmp = u; u = v; v = tmp;
        }
        // (row lo, col lo) = cA, (row lo, col hi) = cH, (row hi, col lo) = cV, (row hi, col hi) = cD
        pass_cols<FMA>(p, out + (size_t)(2 * pl) * hw, out + (size_t)(2 * pl + 1) * hw, H, W, t, 1 << (n - 1));
    }
}

HT_INLINE void load_plane(const void *in, int in_dtype, int in_layout, int b, int c, int C, int H, int W, float *dst)
{
    const size_t hw = (size_t)H * W;
    if (in_dtype == WV_DT_U8) {
        const uint8_t *p = (const uint8_t *)in;
        if (in_layout == WV_LAYOUT_NCHW) {
            p += ((size_t)b * C + c) * hw;
            for (size_t i = 0; i < hw; ++i) dst[i] = (float)p[i] / 255.0f;
        } else {
            p += (size_t)b * hw * C + c;
            for (size_t i = 0; i < hw; ++i) dst[i] = (float)p[i * C] / 255.0f;
        }
    } else {
        const float *p = (const float *)in;
        if (in_layout == WV_LAYOUT_NCHW) {
            memcpy(dst, p + ((size_t)b * C + c) * hw, hw * sizeof(float));
        } else {
            p += (size_t)b * hw * C + c;
            for (size_t i = 0; i < hw; ++i) dst[i] = p[i * C];
        }
    }
}

template <bool FMA>
HT_INLINE void swt_all(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W, int n,
                       const Taps &t)
{
    const size_t hw = (size_t)H * W;
    Scratch s;
    s.a.resize(hw); s.b.resize(hw); s.c.resize(hw); s.d.resize(hw);
    s.pad.resize((size_t)W + (size_t)t.L * (1u << (n - 1)) * 2 + 8);
    std::vector<float> x(hw);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            load_plane(in, in_dtype, in_layout, b, c, C, H, W, x.data());
            swt_plane<FMA>(x.data(), out + ((size_t)b * C + c) * 4 * hw, H, W, n, t, s);
        }
}

__attribute__((target("avx2,fma"))) void swt_all_fma(const void *in, int in_dtype, int in_layout, float *out, int B,
                                                     int C, int H, int W, int n, const Taps &t)
{
    swt_all<true>(in, in_dtype, in_layout, out, B, C, H, W, n, t);
}

void swt_all_generic(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W, int n,
                     const Taps &t)
{
    swt_all<false>(in, in_dtype, in_layout, out, B, C, H, W, n, t);
}

bool cpu_has_fma()
{
    static const int v = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    return v != 0;
}

// ---------------------------------------------------------------------------------- decimated DWT (dwt.hip twin)
inline int dwt_len(int n, int flen) { return (n + flen - 1) / 2; }

inline int sym_index(int i, int n)
{
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

// src [h][w] -> dlo / dhi (either may be null) [ho][wo], the filtered axis halved; same tap order as k_dwt_axis
void dwt_axis(const float *src, float *dlo, float *dhi, int h, int w, int axis, const Taps &t)
{
    const int L = t.L;
    const int ho = axis == 0 ? dwt_len(h, L) : h, wo = axis == 1 ? dwt_len(w, L) : w;
    for (int y = 0; y < ho; ++y)
        for (int x = 0; x < wo; ++x) {
            const int n = axis == 0 ? h : w, o = axis == 0 ? y : x;
            float a = 0.f, d = 0.f;
            for (int j = 0; j < L; ++j) {
                const int idx = sym_index(2 * o + 1 - j, n);
                const float v = axis == 0 ? src[(size_t)idx * w + x] : src[(size_t)y * w + idx];
                a = j == 0 ? t.lo[0] * v : fmaf(t.lo[j], v, a);
                d = j == 0 ? t.hi[0] * v : fmaf(t.hi[j], v, d);
            }
            const size_t off = (size_t)y * wo + x;
            if (dlo) dlo[off] = a;
            if (dhi) dhi[off] = d;
        }
}

int check_common(const void *in, const void *out, int B, int C, int H, int W, int in_dtype, int in_layout,
                 const char *what)
{
    HT_REQUIRE(in && out, "%s: null buffer", what);
    HT_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape B=%d C=%d H=%d W=%d", what, B, C, H, W);
    HT_REQUIRE(in_dtype == WV_DT_U8 || in_dtype == WV_DT_F32, "%s: input dtype %d", what, in_dtype);
    HT_REQUIRE(in_layout == WV_LAYOUT_NCHW || in_layout == WV_LAYOUT_NHWC, "%s: bad layout %d", what, in_layout);
    return WV_OK;
}

}  // namespace

extern "C" int wv_swt2d_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H,
                                    int W, int level, const float *dec_lo, const float *dec_hi, int flen)
{
    if (int rc = check_common(in, out, B, C, H, W, in_dtype, in_layout, "swt (host)")) return rc;
    HT_REQUIRE(dec_lo && dec_hi && flen >= 1 && flen <= 32, "swt (host): 1..32 filter taps, got %d", flen);
    HT_REQUIRE(level >= 1 && level <= 12, "swt (host): level %d out of range", level);
    HT_REQUIRE((H % (1 << level)) == 0 && (W % (1 << level)) == 0,
               "swt: H=%d, W=%d must be multiples of 2^level=%d (PyWavelets raises ValueError here)", H, W, 1 << level);
    Taps t{};
    t.L = flen;
    for (int i = 0; i < flen; ++i) { t.lo[i] = dec_lo[i]; t.hi[i] = dec_hi[i]; }
    try {
        if (cpu_has_fma()) swt_all_fma(in, in_dtype, in_layout, out, B, C, H, W, level, t);
        else swt_all_generic(in, in_dtype, in_layout, out, B, C, H, W, level, t);
    } catch (const std::bad_alloc &) {
        HT_FAIL(WV_ENOMEM, "swt (host): out of memory for the %dx%d scratch planes", H, W);
    }
    return WV_OK;
}

extern "C" int wv_rawstack_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H,
                                       int W, int copies)
{
    if (int rc = check_common(in, out, B, C, H, W, in_dtype, in_layout, "rawstack (host)")) return rc;
    HT_REQUIRE(copies > 0, "rawstack (host): copies must be positive");
    const size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float *dst = out + ((size_t)b * C + c) * copies * hw;
            load_plane(in, in_dtype, in_layout, b, c, C, H, W, dst);
            for (int k = 1; k < copies; ++k) memcpy(dst + (size_t)k * hw, dst, hw * sizeof(float));
        }
    return WV_OK;
}

extern "C" int wv_dwt2d_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H,
                                    int W, int level, const float *dec_lo, const float *dec_hi, int flen)
{
    if (int rc = check_common(in, out, B, C, H, W, in_dtype, in_layout, "dwt (host)")) return rc;
    HT_REQUIRE(dec_lo && dec_hi && flen >= 2 && flen <= 32, "dwt (host): %d taps (supported: 2..32)", flen);
    HT_REQUIRE(level >= 1 && level <= 12, "dwt (host): level %d out of range", level);
    Taps t{};
    t.L = flen;
    for (int i = 0; i < flen; ++i) { t.lo[i] = dec_lo[i]; t.hi[i] = dec_hi[i]; }
    try {
        const size_t slot = (size_t)(H + flen) * (W + flen);
        std::vector<float> cur(slot), ta(slot), td(slot), nxt(slot);
        int hn = H, wn = W;
        for (int l = 0; l < level; ++l) { hn = dwt_len(hn, flen); wn = dwt_len(wn, flen); }
        const size_t band = (size_t)hn * wn;
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < C; ++c) {
                load_plane(in, in_dtype, in_layout, b, c, C, H, W, cur.data());
                float *o = out + ((size_t)b * C + c) * 4 * band;
                int h = H, w = W;
                for (int l = 1; l <= level; ++l) {
                    const int h2 = dwt_len(h, flen);
                    const bool last = l == level;
                    dwt_axis(cur.data(), ta.data(), last ? td.data() : nullptr, h, w, 0, t);
                    if (!last) {
                        dwt_axis(ta.data(), nxt.data(), nullptr, h2, w, 1, t);
                        cur.swap(nxt);
                    } else {
                        dwt_axis(ta.data(), o, o + 2 * band, h2, w, 1, t);         // aa = cA, ad = cV
                        dwt_axis(td.data(), o + band, o + 3 * band, h2, w, 1, t);  // da = cH, dd = cD
                    }
                    h = h2; w = dwt_len(w, flen);
                }
            }
    } catch (const std::bad_alloc &) {
        HT_FAIL(WV_ENOMEM, "dwt (host): out of memory");
    }
    return WV_OK;
}

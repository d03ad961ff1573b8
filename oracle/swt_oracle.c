/*
 * oracle/swt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread, fp32, no FMA contraction) of the 2-D
 * stationary wavelet transform the reference calls at
 *   main/transforms/custom_transforms.py:160-166  (SWTTransform._apply_wavelet ->
 *   pywt.swt2(channel, wavelet, level)[0])
 * and of the image wrapper at
 *   main/transforms/custom_transforms.py:145-157  (BaseWaveletTransform.__call__).
 *
 * The arithmetic itself lives in PyWavelets, a third-party C library that is NOT vendored
 * in the reference, NOT pinned in its requirements.txt and NOT installed in this image.
 * The algorithm restated here is PyWavelets' published one:
 *   swt2 -> swtn: for level l = 1..n, on the running approximation A_{l-1}, for axis 0
 *   then axis 1: swt_axis -> <type>_swt_(level=l) ->
 *   downsampling_convolution_periodization(input, N, e_filter, L*2^(l-1), out, step=1,
 *   fstep=2^(l-1)) with e_filter the zero-stuffed ("a trous") decomposition filter.
 * Its 1-D rule, with s = 2^(l-1) and L taps:
 *   y[o] = sum_{m=0}^{L-1} f[m] * x[(o + s*(L/2 - m)) mod N]          
 * accumulated in tap order m = 0..L-1 in single precision (float input -> float path).
 *
 * PARITY STATUS: "parity unpinned" by PyWavelets itself (it cannot be imported here).
 * What pins this file instead is listed in tests/test_oracle_swt.py: the Haar closed form
 * and the value ranges recorded in studies/results/swt_transform_check_2026-08-12.txt,
 * constant-image gain 2^n, zero-sum detail bands, the 4x energy identity per level for
 * orthonormal filters, shift equivariance, impulse responses, and agreement with the
 * independent numpy restatement oracle/swt_np.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* built with -ffp-contract=off (oracle/Makefile): products round before the add */

/* One periodized a-trous pass (PyWavelets downsampling_convolution_periodization with step = 1, fstep = s) over a
 * whole H x W plane along `axis`: every output element is   sum = 0; for m = 0..L-1: sum = sum + f[m] * x[idx_m]
 * with idx_m = (o + s*(L/2 - m)) mod N -- products rounded before the add, taps in order.  The loops run tap-outer /
 * element-inner over whole rows (each element still sees exactly that sequence of operations), which the compiler can
 * vectorise: the CPU baseline should not be slower than a careful C library would be. */
static long wrap_index(long idx, int n)
{
    idx %= n;
    return idx < 0 ? idx + n : idx;
}

static void atrous_plane(const float *A, int H, int W, const float *f, int L, int s, int axis, float *Y, float *pad)
{
    if (axis == 0) {
        for (int y = 0; y < H; ++y) {
            float *out = Y + (size_t)y * W;
            for (int x = 0; x < W; ++x) out[x] = 0.0f;
            for (int m = 0; m < L; ++m) {
                const float *src = A + (size_t)wrap_index((long)y + (long)s * (L / 2 - m), H) * W;
                const float fm = f[m];
                for (int x = 0; x < W; ++x) {
                    float prod = fm * src[x];
                    out[x] = out[x] + prod;
                }
            }
        }
        return;
    }
    /* axis 1: left / right periodic margins so that the inner loop reads a contiguous window */
    long dmin = 0, dmax = 0;
    for (int m = 0; m < L; ++m) {
        long d = (long)s * (L / 2 - m);
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
    }
    const long left = -dmin, right = dmax;
    for (int y = 0; y < H; ++y) {
        const float *row = A + (size_t)y * W;
        for (long i = 0; i < left; ++i) pad[i] = row[wrap_index(i - left, W)];
        memcpy(pad + left, row, (size_t)W * sizeof(float));
        for (long i = 0; i < right; ++i) pad[left + W + i] = row[wrap_index(i, W)];
        float *out = Y + (size_t)y * W;
        for (int x = 0; x < W; ++x) out[x] = 0.0f;
        for (int m = 0; m < L; ++m) {
            const float *src = pad + left + (long)s * (L / 2 - m);
            const float fm = f[m];
            for (int x = 0; x < W; ++x) {
                float prod = fm * src[x];
                out[x] = out[x] + prod;
            }
        }
    }
}

/* 2-D single-level step on A (H x W, row-major): axis 0 first, then axis 1
 * (PyWavelets swtn iterates `for axis in axes` with axes = (-2, -1)).
 * Band keys: first letter = axis 0.  aa = cA, da = cH, ad = cV, dd = cD. */
static void swt2_level(const float *A, int H, int W, const float *lo, const float *hi, int L,
                       int s, float *aa, float *da, float *ad, float *dd, float *tmp_a,
                       float *tmp_d, float *pad)
{
    atrous_plane(A, H, W, lo, L, s, 0, tmp_a, pad);
    atrous_plane(A, H, W, hi, L, s, 0, tmp_d, pad);
    atrous_plane(tmp_a, H, W, lo, L, s, 1, aa, pad);
    atrous_plane(tmp_a, H, W, hi, L, s, 1, ad, pad);
    atrous_plane(tmp_d, H, W, lo, L, s, 1, da, pad);
    atrous_plane(tmp_d, H, W, hi, L, s, 1, dd, pad);
}

/* pywt.swt2(plane, wavelet, level)[0] -> out[4][H][W] = (cA, cH, cV, cD) of level `level`.
 * Returns 0, or -1 on bad arguments (pywt raises when H or W is not a multiple of
 * 2^level), -2 on allocation failure. */
int wvo_swt2_plane_f32(const float *plane, int H, int W, const float *dec_lo, const float *dec_hi,
                       int L, int level, float *out)
{
    if (!plane || !out || !dec_lo || !dec_hi || H <= 0 || W <= 0 || L <= 0 || level < 1 ||
        level > 16)
        return -1;
    if ((H % (1 << level)) != 0 || (W % (1 << level)) != 0) return -1;
    size_t n = (size_t)H * W;
    float *cur = (float *)malloc(n * sizeof(float));
    float *ta = (float *)malloc(n * sizeof(float));
    float *td = (float *)malloc(n * sizeof(float));
    float *pad = (float *)malloc(((size_t)W + 2 * (size_t)L * ((size_t)1 << (level - 1)) + 8) * sizeof(float));
    if (!cur || !ta || !td || !pad) {
        free(cur); free(ta); free(td); free(pad);
        return -2;
    }
    memcpy(cur, plane, n * sizeof(float));
    float *aa = out, *da = out + n, *ad = out + 2 * n, *dd = out + 3 * n;
    for (int l = 1; l <= level; ++l) {
        swt2_level(cur, H, W, dec_lo, dec_hi, L, 1 << (l - 1), aa, da, ad, dd, ta, td, pad);
        if (l < level) memcpy(cur, aa, n * sizeof(float));
    }
    free(cur); free(ta); free(td); free(pad);
    return 0;
}

/* BaseWaveletTransform.__call__ on one already-sized image: HWC uint8 -> [3][4][H][W] f32,
 * x/255 in fp32 (numpy `astype(float32) / 255.0`), per channel, stacked channel-major.
 * mode 0 = SWT, mode 1 = RawStack (4 copies; custom_transforms.py:184-185). */
int wvo_transform_image_u8(const uint8_t *hwc, int H, int W, const float *dec_lo,
                           const float *dec_hi, int L, int level, int mode, float *out)
{
    if (!hwc || !out || H <= 0 || W <= 0) return -1;
    size_t n = (size_t)H * W;
    float *plane = (float *)malloc(n * sizeof(float));
    if (!plane) return -2;
    int rc = 0;
    for (int c = 0; c < 3 && rc == 0; ++c) {
        for (size_t i = 0; i < n; ++i) plane[i] = (float)hwc[i * 3 + c] / 255.0f;
        float *o = out + (size_t)c * 4 * n;
        if (mode == 1) {
            for (int b = 0; b < 4; ++b) memcpy(o + b * n, plane, n * sizeof(float));
        } else {
            rc = wvo_swt2_plane_f32(plane, H, W, dec_lo, dec_hi, L, level, o);
        }
    }
    free(plane);
    return rc;
}

/* Batched driver used by the CPU baseline: images [B][H][W][3] u8 -> [B][3][4][H][W] f32,
 * one image at a time exactly as the reference's DataLoader worker does. */
int wvo_transform_batch_u8(const uint8_t *imgs, int B, int H, int W, const float *dec_lo,
                           const float *dec_hi, int L, int level, int mode, float *out)
{
    size_t in_sz = (size_t)H * W * 3, out_sz = (size_t)H * W * 12;
    for (int b = 0; b < B; ++b) {
        int rc = wvo_transform_image_u8(imgs + b * in_sz, H, W, dec_lo, dec_hi, L, level, mode,
                                        out + b * out_sz);
        if (rc) return rc;
    }
    return 0;
}

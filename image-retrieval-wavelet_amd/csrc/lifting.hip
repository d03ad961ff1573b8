// Legacy lifting-scheme DWT (CustomTransform: HaarLifting / Cdf97Lifting) for gfx950 -- completeness of the
// transform plugin family, plain kernels outside the hot path.
//
// Reference: /root/reference/main/transforms/wavelets/haar.py:21-43,69-86, cdf_97.py:33-73,119-133,
// utils.py:376-392,401-460 and the wrappers at custom_transforms.py:14-55.  One level = a 1-D lifting pass along
// H (even / odd ROWS), then along W, both with zero-padded shifts ('constant' pad of the shifted operand at every
// lifting step, not a signal extension), coefficients joined as [s | d] per axis; LL = top-left, LH = bottom-left,
// HL = top-right, HH = bottom-right, then the 2-D scales (1/sqrt(2)^2, 1, 1, sqrt(2)).
// Every product and sum is rounded separately and in the reference's order (__fmul_rn / __fadd_rn: no FMA
// contraction), so the output is bit-identical to the reference's float32 torch ops.
#include "common.hpp"
#include <cmath>

// hipcc contracts a*b + c into an FMA by default, and its __fmul_rn / __fadd_rn are plain operators that take part in
// that: this file is compiled with -ffp-contract=off (csrc/Makefile), which is what makes the result bit-identical to
// the reference.  The pragma covers builds that forget the flag where the compiler honours it.
#pragma clang fp contract(off)

namespace wv {

struct LiftConsts {
    float a1, a2, a3, a4, k, rk;    // cdf 9/7 steps and scale; for haar k = sqrt(2), rk = 1/sqrt(2)
    float sc_ll, sc_hh;             // 2-D scales of LL and HH (LH, HL: 1)
};

// s and d of pair index i along a strided 1-D signal of n pairs: ev(j) = p[2j*stride], od(j) = p[(2j+1)*stride]
template <int BASIS>
__device__ __forceinline__ void lift_pair(const float *__restrict__ p, int64_t stride, int n, int i, const LiftConsts &c,
                                          float &s, float &d)
{
    auto ev = [&](int j) { return p[(int64_t)(2 * j) * stride]; };
    auto od = [&](int j) { return p[(int64_t)(2 * j + 1) * stride]; };
    if constexpr (BASIS == 0) {
        const float e = ev(i);
        const float od1 = __fadd_rn(od(i), __fmul_rn(-1.0f, e));
        const float ev1 = __fadd_rn(e, __fmul_rn(0.5f, od1));
        s = __fmul_rn(c.k, ev1);
        d = __fmul_rn(c.rk, od1);
    } else {
        // od1 on [i-2, i+1], ev1 on [i-1, i+1], od2 on [i-1, i], ev2 at i; an index outside [0, n) stands for the zero pad
        float e[5];                       // ev(i-2 .. i+2)
#pragma unroll
        for (int u = 0; u < 5; ++u) { const int j = i - 2 + u; e[u] = (j >= 0 && j < n) ? ev(j) : 0.f; }
        float od1[4];                     // od1(i-2 .. i+1)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i - 2 + u;
            od1[u] = (j >= 0 && j < n) ? __fadd_rn(od(j), __fadd_rn(__fmul_rn(c.a1, e[u]), __fmul_rn(c.a1, e[u + 1]))) : 0.f;
        }
        float ev1[3];                     // ev1(i-1 .. i+1)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int j = i - 1 + u;
            ev1[u] = (j >= 0 && j < n) ? __fadd_rn(e[u + 1], __fadd_rn(__fmul_rn(c.a2, od1[u]), __fmul_rn(c.a2, od1[u + 1]))) : 0.f;
        }
        float od2[2];                     // od2(i-1, i)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = i - 1 + u;
            od2[u] = (j >= 0 && j < n) ? __fadd_rn(od1[u + 1], __fadd_rn(__fmul_rn(c.a3, ev1[u]), __fmul_rn(c.a3, ev1[u + 1]))) : 0.f;
        }
        const float ev2 = __fadd_rn(ev1[1], __fadd_rn(__fmul_rn(c.a4, od2[0]), __fmul_rn(c.a4, od2[1])));
        s = __fmul_rn(c.k, ev2);
        d = __fmul_rn(c.rk, od2[1]);
    }
}

// pass along H: in [P][H][W] -> tmp [P][H][W], rows [0, H/2) = s, [H/2, H) = d
template <int BASIS>
__global__ void k_lift_rows(const float *__restrict__ in, float *__restrict__ tmp, int64_t P, int H, int W, LiftConsts c)
{
    const int n = H / 2;
    const int64_t total = P * n * W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), i = (int)((t / W) % n);
        const int64_t p = t / ((int64_t)W * n);
        float s, d;
        lift_pair<BASIS>(in + p * H * W + x, W, n, i, c, s, d);
        tmp[(p * H + i) * W + x] = s;
        tmp[(p * H + n + i) * W + x] = d;
    }
}

// pass along W + band split + 2-D scales: tmp -> ll [P][H/2][W/2], hi [P][3][H/2][W/2] (LH, HL, HH)
template <int BASIS>
__global__ void k_lift_cols(const float *__restrict__ tmp, float *__restrict__ ll, float *__restrict__ hi, int64_t P, int H,
                            int W, LiftConsts c)
{
    const int n = W / 2, hh = H / 2;
    const int64_t total = P * H * n;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % n), y = (int)((t / n) % H);
        const int64_t p = t / ((int64_t)n * H);
        float s, d;
        lift_pair<BASIS>(tmp + (p * H + y) * W, 1, n, i, c, s, d);
        const int64_t band = (int64_t)hh * n;
        if (y < hh) {      // rows s: LL (cols s), HL (cols d)
            ll[p * band + (int64_t)y * n + i] = __fmul_rn(s, c.sc_ll);
            hi[(p * 3 + 1) * band + (int64_t)y * n + i] = d;
        } else {           // rows d: LH (cols s), HH (cols d)
            hi[(p * 3 + 0) * band + (int64_t)(y - hh) * n + i] = s;
            hi[(p * 3 + 2) * band + (int64_t)(y - hh) * n + i] = __fmul_rn(d, c.sc_hh);
        }
    }
}

}  // namespace wv

using namespace wv;

extern "C" size_t wv_lifting2d_workspace_bytes(int64_t planes, int H, int W)
{
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)planes * H * W * sizeof(float);
}

extern "C" int wv_lifting2d_forward(const float *in, int64_t planes, int H, int W, int basis, float *ll, float *hi,
                                    void *workspace, size_t workspace_bytes, void *stream)
{
    WV_REQUIRE(in && ll && hi, "lifting2d: null buffer");
    WV_REQUIRE(planes >= 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "lifting2d: H=%d W=%d must be even (pad first)", H, W);
    WV_REQUIRE(basis == 0 || basis == 1, "lifting2d: basis %d (0 = haar, 1 = cdf97)", basis);
    if (planes == 0) return WV_OK;
    const size_t need = wv_lifting2d_workspace_bytes(planes, H, W);
    if (!workspace || workspace_bytes < need) WV_FAIL(WV_ENOMEM, "lifting2d: workspace %zu < %zu bytes", workspace_bytes, need);
    LiftConsts c;
    if (basis == 0) {
        const double k = std::sqrt(2.0);
        c = {0.f, 0.f, 0.f, 0.f, (float)k, (float)(1.0 / k), 0.f, 0.f};
    } else {
        const double k = 1.149604398;
        c = {(float)-1.58613432, (float)-0.05298011854, (float)0.8829110762, (float)0.4435068522, (float)k, (float)(1.0 / k), 0.f, 0.f};
    }
    c.sc_ll = (float)(1.0 / (std::sqrt(2.0) * std::sqrt(2.0)));
    c.sc_hh = (float)std::sqrt(2.0);
    hipStream_t st = (hipStream_t)stream;
    float *tmp = (float *)workspace;
    const int64_t na = planes * (H / 2) * W, nb = planes * H * (W / 2);
    const unsigned ga = (unsigned)std::min<int64_t>((na + 255) / 256, 1 << 16), gb = (unsigned)std::min<int64_t>((nb + 255) / 256, 1 << 16);
    if (basis == 0) {
        hipLaunchKernelGGL((k_lift_rows<0>), dim3(ga), dim3(256), 0, st, in, tmp, planes, H, W, c);
        hipLaunchKernelGGL((k_lift_cols<0>), dim3(gb), dim3(256), 0, st, tmp, ll, hi, planes, H, W, c);
    } else {
        hipLaunchKernelGGL((k_lift_rows<1>), dim3(ga), dim3(256), 0, st, in, tmp, planes, H, W, c);
        hipLaunchKernelGGL((k_lift_cols<1>), dim3(gb), dim3(256), 0, st, tmp, ll, hi, planes, H, W, c);
    }
    WV_CHECK_LAUNCH("lifting2d");
    return WV_OK;
}

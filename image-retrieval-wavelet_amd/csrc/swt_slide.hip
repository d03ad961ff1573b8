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
#include <vector>

// Folding the u8 -> [0,1] division into the last row-filter taps saves 3 VALU ops per pixel but measured
// SLOWER on MI355X (1.265 ms vs 1.180 ms, A/B in one session, db2 level 3): off.
#ifndef WV_SWT_FOLD255
#define WV_SWT_FOLD255 0
#endif

// A/B on MI355X in one session (tools/build_variant.sh), db2 level 3, 2048 images:
//   consumer (column) waves at raised issue priority: 1.148 ms vs 1.19 ms at equal priority (they are the critical role)
//   interior chunks storing without per-row bounds checks (a second copy of the store loop): 1.18 ms, slower -> off
#ifndef WV_SWT_VPRIO
#define WV_SWT_VPRIO 2
#endif
#ifndef WV_SWT_FASTMID
#define WV_SWT_FASTMID 0
#endif
// Column pass on (row-lo, row-hi) PAIRS: the two row-filtered planes go through identical arithmetic, so the
// consumer keeps them as 2-vectors and every multiply-add becomes one v_pk_fma_f32 -- half the instructions a
// wave has to issue for the same (bitwise identical) result.  The LDS ring then holds the planes interleaved.
#ifndef WV_SWT_PAIRED
#define WV_SWT_PAIRED 1
#endif

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
    int coal;       // planar uint8, W % 16 == 0: producers load every pixel once, coalesced (see the producer role)
    int nxcd;       // 8: workgroups w, w+8, ... (one XCD, hardware round-robin) share images; 1: plain striding
    // output addressing in elements: plane (b, c) starts at (b*C + c) * pstride, its band k at + k * bstride.
    //   [B][C][4][H][W] (reference layout):  pstride = 4*H*W, bstride = H*W
    //   [4][B][C][H][W] (band-major, each band one contiguous NCHW batch):  pstride = H*W, bstride = B*C*H*W
    size_t pstride, bstride;
    unsigned long long *stamps;   // diagnostic build only
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

// Streaming form of the same cascade for pass V: a thread keeps, per column, the last (L-1)*2^(l-1)
// inputs of every level l in registers ("tails", HALO values in all), so each step costs exactly
// L MACs per level per new row -- no halo recompute -- and emits outputs delayed by HALO rows.
// The arithmetic per output is identical to SChain (same taps, same order).
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float fma_s(float s, float x, float a) { return fmaf(s, x, a); }
__device__ __forceinline__ f32x2 fma_s(float s, f32x2 x, f32x2 a)
{
    const f32x2 sv = {s, s};
    return __builtin_elementwise_fma(sv, x, a);
}

template <int L, int NLEV, int N, typename T = float>
struct VStep {
    static constexpr int HALO = (L - 1) * ((1 << NLEV) - 1);
    // levels LEV .. NLEV-1: cur[] holds N new level-LEV inputs on entry, N new level-NLEV inputs on exit
    template <int LEV>
    static __device__ __forceinline__ void lower(T (&cur)[N], T (&tail)[HALO], const float (&lo)[L])
    {
        if constexpr (LEV < NLEV) {
            constexpr int S = 1 << (LEV - 1), TT = (L - 1) * S, OFF = (L - 1) * (S - 1);
            T seq[TT + N];
#pragma unroll
            for (int i = 0; i < TT; ++i) seq[i] = tail[OFF + i];
#pragma unroll
            for (int i = 0; i < N; ++i) seq[TT + i] = cur[i];
#pragma unroll
            for (int t = 0; t < N; ++t) {
                T a = lo[0] * seq[t + S * (L - 1)];
#pragma unroll
                for (int m = 1; m < L; ++m) a = fma_s(lo[m], seq[t + S * (L - 1 - m)], a);
                cur[t] = a;
            }
#pragma unroll
            for (int i = 0; i < TT; ++i) tail[OFF + i] = seq[N + i];
            lower<LEV + 1>(cur, tail, lo);
        }
    }
    // last level: seq = tail ++ cur; out(t) for t < N; tail updated
    template <typename Emit>
    static __device__ __forceinline__ void last(const T (&cur)[N], T (&tail)[HALO], const float (&lo)[L],
                                                const float (&hi)[L], Emit emit)
    {
        constexpr int S = 1 << (NLEV - 1), TT = (L - 1) * S, OFF = (L - 1) * (S - 1);
        T seq[TT + N];
#pragma unroll
        for (int i = 0; i < TT; ++i) seq[i] = tail[OFF + i];
#pragma unroll
        for (int i = 0; i < N; ++i) seq[TT + i] = cur[i];
#pragma unroll
        for (int t = 0; t < N; ++t) {
            T a = lo[0] * seq[t + S * (L - 1)], d = hi[0] * seq[t + S * (L - 1)];
#pragma unroll
            for (int m = 1; m < L; ++m) {
                a = fma_s(lo[m], seq[t + S * (L - 1 - m)], a);
                d = fma_s(hi[m], seq[t + S * (L - 1 - m)], d);
            }
            emit(t, a, d);
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) tail[OFF + i] = seq[N + i];
    }
    // warm-up: only refresh the last level's tail
    static __device__ __forceinline__ void prime_last(const T (&cur)[N], T (&tail)[HALO])
    {
        constexpr int S = 1 << (NLEV - 1), TT = (L - 1) * S, OFF = (L - 1) * (S - 1);
        T seq[TT + N];
#pragma unroll
        for (int i = 0; i < TT; ++i) seq[i] = tail[OFF + i];
#pragma unroll
        for (int i = 0; i < N; ++i) seq[TT + i] = cur[i];
#pragma unroll
        for (int i = 0; i < TT; ++i) tail[OFF + i] = seq[N + i];
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

// FOLD = true: leave the pixels as integers 0..255 (one v_cvt each); the 1/255 is folded into the taps of the
// last row-filter level by the caller (the transform is linear).
template <typename InT, int LAYOUT, bool FOLD = false>
__device__ __forceinline__ float4 s_convert4(const SRaw<InT, LAYOUT> &r, int c)
{
    if constexpr (sizeof(InT) == 1 && FOLD) {
        if constexpr (LAYOUT == 1) {
            const uint32_t s0 = __builtin_amdgcn_alignbyte(r.d[1], r.d[0], (uint32_t)c);
            const uint32_t s1 = __builtin_amdgcn_alignbyte(r.d[2], r.d[1], (uint32_t)c);
            const uint32_t s2 = __builtin_amdgcn_alignbyte(0u, r.d[2], (uint32_t)c);
            return make_float4(s_ubyte<0>(s0), s_ubyte<3>(s0), s_ubyte<2>(s1), s_ubyte<1>(s2));
        } else {
            const uint32_t d = r.d[0];
            return make_float4(s_ubyte<0>(d), s_ubyte<1>(d), s_ubyte<2>(d), s_ubyte<3>(d));
        }
    } else if constexpr (sizeof(InT) == 1) {
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

// Producer / consumer split inside one workgroup of 2 * NH threads:
//   waves [NH/64, 2*NH/64)  "H waves": fetch the TH input rows of chunk k+1, filter them along x in
//                           registers (all levels) and write lo / hi into LDS buffer (k+1) & 1
//   waves [0, NH/64)        "V waves": thread = column; stream chunk k of both planes from LDS buffer
//                           k & 1 through the per-column cascade (tails in registers) and store the TH
//                           output rows the cascade emits (it lags the input by HALO rows)
// One barrier per chunk hands the buffers over.  Each role needs well under 128 VGPRs, so two
// workgroups (16 waves) fit a CU and the two halves of every SIMD pair overlap load latency,
// row arithmetic, column arithmetic and the output stream.
// STAMP = true is a diagnostic build (WV_SWT_STAMPS=1): per-wave s_memtime totals of the three segments of
// each role go to g.stamps (never read by the kernel, never part of an output).
#define WV_STAMP(var) do { if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
// Store through "SGPR base + 32-bit lane offset" addressing (global_store ... saddr): the row pointer is
// wave-uniform, so no per-store 64-bit vector address arithmetic is needed.  hipcc builds the address in
// VGPRs for this pattern, hence the explicit instruction.  Only V waves store and they issue no vector
// loads in their loop, so the compiler's vmcnt bookkeeping is unaffected.
__device__ __forceinline__ void store_row(float *row_uniform, uint32_t byte_off, float v)
{
#ifdef WV_SWT_NOSTORE   // diagnostic build: arithmetic only (the store survives only for a value that never occurs)
    if (__float_as_uint(v) == byte_off * 0x9E3779B1u + 0x7fc12345u)
        *reinterpret_cast<float *>(reinterpret_cast<char *>(row_uniform) + byte_off) = v;
    return;
#endif
    asm volatile("global_store_dword %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(row_uniform) : "memory");
}
__device__ __forceinline__ void store_row(__hip_bfloat16 *row_uniform, uint32_t byte_off, __hip_bfloat16 v)
{
    *reinterpret_cast<__hip_bfloat16 *>(reinterpret_cast<char *>(row_uniform) + byte_off) = v;
}

template <int L, int NLEV, int R, int TH, int NH, int MINW, typename InT, int LAYOUT, bool BF16, bool STAMP = false>
__global__ __launch_bounds__(2 * NH, MINW) void k_swt_slide(const InT *__restrict__ in, void *__restrict__ out,
                                                            SlideGeom g, STaps<L> taps)
{
    uint64_t st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, acc_a = 0, acc_b = 0, acc_c = 0, acc_d = 0, acc_e = 0;
    using CH = SChain<L, NLEV, R>;
    using Raw = SRaw<InT, LAYOUT>;
    using OutT = typename std::conditional<BF16, __hip_bfloat16, float>::type;
    constexpr int HALO = CH::HALO;
    constexpr int HB = (L / 2 - 1) * ((1 << NLEV) - 1);
    constexpr int HBa = (HB + 3) / 4 * 4;
    constexpr int NG = (HBa - HB + CH::NIN + 3) / 4;      // 4-pixel groups one run loads
    static_assert(R % 4 == 0, "run length must be a multiple of 4");
    extern __shared__ float4 ring4[];
    float *ring = reinterpret_cast<float *>(ring4);       // [2 buffers][2 planes][TH][P]
    const int P = g.P, W = g.W, H = g.H;
    const int plane_sz = TH * P, buf_sz = 2 * plane_sz;
    const int nchunks = (H + HALO + TH - 1) / TH;         // the cascade consumes H + HALO rows per plane
    const uint32_t band = (uint32_t)H * W;
    // Plane schedule.  Workgroup w runs on XCD w % 8 (round-robin dispatch) and every XCD has its own L2, so
    // the C channel planes of one interleaved (NHWC) image -- which all read the same bytes -- go to
    // neighbouring workgroups of ONE XCD: image b belongs to XCD b % nxcd, and the planes of an XCD's images
    // are dealt to its workgroups in (image, channel) order.  Two of the three reads then hit that L2.
    const int xcd = blockIdx.x % g.nxcd, wg_in_xcd = blockIdx.x / g.nxcd, wgs_per_xcd = gridDim.x / g.nxcd;
    const int nq = g.B > xcd ? (g.B - xcd + g.nxcd - 1) / g.nxcd * g.C : 0;   // planes this XCD owns
    const bool is_h = threadIdx.x >= NH;                  // wave-uniform role
    const int t = is_h ? threadIdx.x - NH : threadIdx.x;

    constexpr bool FOLD = sizeof(InT) == 1 && WV_SWT_FOLD255;
    constexpr bool PAIRED = WV_SWT_PAIRED != 0;
    float hlo[L], hhi[L];   // taps of the last row-filter level (scaled by 1/255 when the division is folded)
#pragma unroll
    for (int m = 0; m < L; ++m) {
        hlo[m] = FOLD ? taps.lo[m] * 0.003921568859368563f : taps.lo[m];
        hhi[m] = FOLD ? taps.hi[m] * 0.003921568859368563f : taps.hi[m];
    }
    if (is_h) {
        // ------------------------------------------------------------------ producer: pass H
        // (row, run) of this producer thread.  Coalesced mode (planar uint8, W % 16 == 0) keeps the runs of one row
        // inside one wave -- 64 / nrun rows per wave -- so that neighbouring runs can hand over their pixels with
        // ds_bpermute: each thread then issues ONE 16-byte load per chunk and every input byte is read exactly once
        // (the strided form touches each cache line of its 40-pixel window ten times).
        int rr = t / g.nrun, j = t - rr * g.nrun;
        bool active = t < TH * g.nrun;
        int lane_left = 0, lane_right = 0, rr_ld = 0;
        if (g.coal) {
            const int l = t & 63, rpw = 64 / g.nrun, lr = l / g.nrun;
            j = l - lr * g.nrun;
            rr = (t >> 6) * rpw + lr;
            active = lr < rpw && rr < TH;
            rr_ld = min(rr, TH - 1);                                             // spare lanes load a valid row too
            lane_left = (lr * g.nrun + (j == 0 ? g.nrun - 1 : j - 1)) * 4;       // byte addresses for ds_bpermute
            lane_right = (lr * g.nrun + (j == g.nrun - 1 ? 0 : j + 1)) * 4;      // (periodic: the row wraps onto itself)
            if (lr >= rpw) lane_left = lane_right = 0;
        }
        for (int q = wg_in_xcd; q < nq; q += wgs_per_xcd) {
            const int m = q / g.C, c = q - m * g.C, b = xcd + g.nxcd * m, pc = b * g.C + c;
            const InT *img = LAYOUT == 0 ? in + (size_t)pc * band : in + (size_t)b * band * 3;
            Raw raw[NG];
            auto fetch = [&](int chunk) {
                // input row of virtual row v = chunk*TH + rr is y = v - HB (wrapped; rows past H + HA wrap too)
                int y = chunk * TH + rr - HB;
                y = y >= H ? y - H : y;
                const uint32_t row = (uint32_t)swrap1(y, H) * (uint32_t)W;
                const int gx0 = j * R - HBa;
#pragma unroll
                for (int k = 0; k < NG; ++k)
                    raw[k] = s_fetch4<InT, LAYOUT>(img, row + (uint32_t)swrap1(gx0 + 4 * k, W), c);
            };
            constexpr bool CAN_COAL = sizeof(InT) == 1 && LAYOUT == 0 && R == 16 && NG * 4 - HBa <= 32;
            uint32_t own[4] = {0, 0, 0, 0};
            auto issue_row = [&](int chunk) {       // every lane of the wave (the exchange below reads all of them)
                int y = chunk * TH + rr_ld - HB;
                y = y >= H ? y - H : y;
                const uint4 v4 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(img) +
                                                                  (size_t)swrap1(y, H) * W + j * R);
                own[0] = v4.x; own[1] = v4.y; own[2] = v4.z; own[3] = v4.w;
            };
            const bool coal = CAN_COAL && g.coal;   // uniform
            if (coal) issue_row(0);
            else if (active) fetch(0);
            for (int k = 0; k < nchunks; ++k) {
                WV_STAMP(st0);
                if constexpr (CAN_COAL) {
                    if (coal) {
                        // window of run j = pixels [16j - HBa, 16j - HBa + 4 NG): left neighbour | own | right neighbour
#pragma unroll
                        for (int kk = 0; kk < NG; ++kk) {
                            const int gpx = 4 * kk - HBa;                      // compile-time after unrolling
                            if (gpx < 0)
                                raw[kk].d[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_left, (int)own[(gpx + 16) / 4]);
                            else if (gpx < 16)
                                raw[kk].d[0] = own[gpx / 4];
                            else
                                raw[kk].d[0] = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_right, (int)own[(gpx - 16) / 4]);
                        }
                        if (k + 1 < nchunks) issue_row(k + 1);                 // flies during this chunk's arithmetic
                    }
                }
                if (active) {
                    float v[NG * 4];
#pragma unroll
                    for (int q = 0; q < NG; ++q) {
                        const float4 p4 = s_convert4<InT, LAYOUT, FOLD>(raw[q], c);
                        v[4 * q + 0] = p4.x; v[4 * q + 1] = p4.y; v[4 * q + 2] = p4.z; v[4 * q + 3] = p4.w;
                    }
                    WV_STAMP(st1);
                    if (k + 1 < nchunks && !coal) fetch(k + 1);     // next chunk's pixels fly during the arithmetic
                    WV_STAMP(st4);
                    // in-place cascade on v[off ..): element i of the run lives at v[i + off]
                    constexpr int off = HBa - HB;
                    float w[CH::NIN];
#pragma unroll
                    for (int i = 0; i < CH::NIN; ++i) w[i] = v[i + off];
                    CH::template lower<1>(w, taps.lo);
                    WV_STAMP(st5);
                    // LDS row layout is permuted so that consecutive lanes (= consecutive runs j) write
                    // consecutive 16-byte slots: column x = j*R + 4*q4 + e lives at q4*(4*nrun) + 4*j + e
                    if constexpr (PAIRED) {
                        // interleaved planes: column x = j*R + 2*q2 + e holds (lo, hi) at floats
                        // q2*(4*nrun) + 4*j + 2*e of its row -> lanes (= runs j) write consecutive 16-byte slots
                        float *prow = ring + (k & 1) * buf_sz + rr * (2 * P) + 4 * j;
                        const int qstride = 4 * g.nrun;
#pragma unroll
                        for (int q2 = 0; q2 < R / 2; ++q2) {
                            float4 v4;
                            v4.x = CH::last(w, hlo, 2 * q2 + 0); v4.y = CH::last(w, hhi, 2 * q2 + 0);
                            v4.z = CH::last(w, hlo, 2 * q2 + 1); v4.w = CH::last(w, hhi, 2 * q2 + 1);
                            *reinterpret_cast<float4 *>(prow + q2 * qstride) = v4;
                        }
                    } else {
                    float *plo = ring + (k & 1) * buf_sz + rr * P + 4 * j;
                    float *phi = plo + plane_sz;
                    const int qstride = 4 * g.nrun;
#pragma unroll
                    for (int q4 = 0; q4 < R / 4; ++q4) {
                        float4 lo4, hi4;
                        lo4.x = CH::last(w, hlo, 4 * q4 + 0); hi4.x = CH::last(w, hhi, 4 * q4 + 0);
                        lo4.y = CH::last(w, hlo, 4 * q4 + 1); hi4.y = CH::last(w, hhi, 4 * q4 + 1);
                        lo4.z = CH::last(w, hlo, 4 * q4 + 2); hi4.z = CH::last(w, hhi, 4 * q4 + 2);
                        lo4.w = CH::last(w, hlo, 4 * q4 + 3); hi4.w = CH::last(w, hhi, 4 * q4 + 3);
                        *reinterpret_cast<float4 *>(plo + q4 * qstride) = lo4;
                        *reinterpret_cast<float4 *>(phi + q4 * qstride) = hi4;
                    }
                    }
                }
                WV_STAMP(st2);
                __syncthreads();   // buffer k & 1 is full; buffer (k+1) & 1 was drained before this barrier
                WV_STAMP(st3);
                if constexpr (STAMP) { acc_a += st1 - st0; acc_b += st2 - st5; acc_c += st3 - st2; acc_d += st4 - st1; acc_e += st5 - st4; }
            }
            __syncthreads();       // pairs with the consumer's last barrier of the plane
        }
    } else {
        // ------------------------------------------------------------------ consumer: pass V
        const bool active = t < W;
        if (WV_SWT_VPRIO) __builtin_amdgcn_s_setprio(WV_SWT_VPRIO);   // the critical role issues first
        // this column's slot in the permuted LDS row (see pass H)
        const int tp = ((t % R) / 4) * (4 * g.nrun) + (t / R) * 4 + (t & 3);
        const int tp2 = ((t % R) / 2) * (4 * g.nrun) + (t / R) * 4 + 2 * (t & 1);   // PAIRED layout (floats)
        for (int q = wg_in_xcd; q < nq; q += wgs_per_xcd) {
            const int m = q / g.C, pc = (xcd + g.nxcd * m) * g.C + (q - m * g.C);
            OutT *oplane = reinterpret_cast<OutT *>(out) + (size_t)pc * g.pstride;
            float tail[PAIRED ? 1 : 2][PAIRED ? 1 : HALO];
            f32x2 tail2[PAIRED ? HALO : 1];
            if constexpr (PAIRED) {
#pragma unroll
                for (int i = 0; i < HALO; ++i) tail2[i] = f32x2{0.f, 0.f};
            } else {
#pragma unroll
                for (int i = 0; i < HALO; ++i) tail[0][i] = tail[1][i] = 0.f;
            }
            __syncthreads();       // chunk 0 produced
            for (int k = 0; k < nchunks; ++k) {
                const int y0 = k * TH - HALO;              // first output row this chunk emits (uniform)
                WV_STAMP(st0);
                if constexpr (PAIRED) {
                    if (active) {
                        using VS = VStep<L, NLEV, TH, f32x2>;
                        const float *col = ring + (k & 1) * buf_sz + tp2;
                        f32x2 cur[TH];
#pragma unroll
                        for (int i = 0; i < TH; ++i) cur[i] = *reinterpret_cast<const f32x2 *>(col + i * (2 * P));
                        VS::template lower<1>(cur, tail2, taps.lo);
                        WV_STAMP(st1);
                        if (y0 + TH <= 0) {
                            VS::prime_last(cur, tail2);
                        } else {
                            // byte offsets of this lane inside the plane's 4-band block (< 2^32, host check)
                            // the four band rows are wave-uniform pointers (SGPR pairs, scalar adds): the band stride
                            // may exceed 32 bits in the band-major layout
                            const uint32_t o0 = (uint32_t)t * (uint32_t)sizeof(OutT);
                            const size_t bs = g.bstride;
                            VS::last(cur, tail2, taps.lo, taps.hi, [&](int i, f32x2 a, f32x2 d) {
                                const int y = y0 + i;
                                if (y >= 0 && y < H) {
                                    OutT *orow = oplane + (size_t)y * W;   // uniform: lives in an SGPR pair
                                    store_row(orow, o0, (OutT)a.x);             // cA  = (row lo, col lo)
                                    store_row(orow + bs, o0, (OutT)d.x);        // cH  = (row lo, col hi)
                                    store_row(orow + 2 * bs, o0, (OutT)a.y);    // cV  = (row hi, col lo)
                                    store_row(orow + 3 * bs, o0, (OutT)d.y);    // cD  = (row hi, col hi)
                                }
                            });
                        }
                    }
                } else
                if (active) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const float *col = ring + (k & 1) * buf_sz + pl * plane_sz + tp;
                        float cur[TH];
#pragma unroll
                        for (int i = 0; i < TH; ++i) cur[i] = col[i * P];
                        VStep<L, NLEV, TH>::template lower<1>(cur, tail[pl], taps.lo);
                        if (pl == 0) WV_STAMP(st1);
                        if (y0 + TH <= 0) {               // nothing to emit yet: only advance the state
                            VStep<L, NLEV, TH>::prime_last(cur, tail[pl]);
                        } else {
                            // byte offsets of this lane inside the plane's 4-band block (< 2^32, host check)
                            const uint32_t off_lo = (uint32_t)t * (uint32_t)sizeof(OutT), off_hi = off_lo;
                            OutT *oplane_lo = oplane + (size_t)(2 * pl) * g.bstride, *oplane_hi = oplane_lo + g.bstride;
                            const size_t hi_delta = g.bstride;
                            (void)oplane_hi;
                            if (WV_SWT_FASTMID && y0 >= 0 && y0 + TH <= H) {   // interior chunk: all TH rows exist
                                OutT *orow0 = oplane_lo + (size_t)y0 * W;
                                VStep<L, NLEV, TH>::last(cur, tail[pl], taps.lo, taps.hi, [&](int i, float a, float d) {
                                    OutT *orow = orow0 + (size_t)i * W;
                                    store_row(orow, off_lo, (OutT)a);
                                    store_row(orow + hi_delta, off_hi, (OutT)d);
                                });
                            } else {
                                VStep<L, NLEV, TH>::last(cur, tail[pl], taps.lo, taps.hi, [&](int i, float a, float d) {
                                    const int y = y0 + i;
                                    if (y >= 0 && y < H) {
                                        OutT *orow = oplane_lo + (size_t)y * W;   // uniform: lives in an SGPR pair
                                        store_row(orow, off_lo, (OutT)a);
                                        store_row(orow + hi_delta, off_hi, (OutT)d);
                                    }
                                });
                            }
                        }
                    }
                }
                WV_STAMP(st2);
                __syncthreads();   // buffer k & 1 drained, buffer (k+1) & 1 full
                WV_STAMP(st3);
                if constexpr (STAMP) { acc_a += st1 - st0; acc_b += st2 - st1; acc_c += st3 - st2; }
            }
        }
    }
    if constexpr (STAMP) {
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *o = g.stamps + ((size_t)blockIdx.x * (2 * NH / 64) + threadIdx.x / 64) * 8;
            o[0] = acc_a; o[1] = acc_b; o[2] = acc_c; o[3] = is_h; o[4] = acc_d; o[5] = acc_e;
        }
    }
}

// Round 3, measured and removed: pulling a workgroup's NEXT plane into L2 in one burst at the start of each plane (one dword per
// 128-byte line, so that HBM sees one 50 KB read per plane instead of fifteen 3.5 KB reads mixed into the write stream):
// 1.126-1.130 ms against 1.062 ms back to back (the write stream evicts the lines before they are used: the reads happen
// twice).  LDS ring pitch W + {0, 4, 8, 12, 28}: no difference (profiles/r03_swt_experiments.txt).
// Decoupled roles were built and measured in round 2 (per-stage LDS ready / free counters instead of the chunk barrier:
// 4 stages x 8 rows with producer wave pairs alternating stages, and 2 stages x 16 rows with the stage released as soon as
// the consumers hold its rows in registers): bit-identical output, 1.22 ms and 1.16 ms against 1.08 ms for this kernel
// (2048 images, back to back) -- polling waves and the shorter per-stage batches cost more than the barrier wait they
// remove, which the second workgroup of the CU was already filling.  Removed; see DESIGN.md section 5.

template <int L, int NLEV, int R, int TH, int NT, int MINW, typename InT, int LAYOUT, bool BF16>
static int launch_slide(const void *in, void *out, SlideGeom g, const float *lo, const float *hi, hipStream_t st)
{
    constexpr int HALO = (L - 1) * ((1 << NLEV) - 1);
    STaps<L> taps;
    for (int i = 0; i < L; ++i) { taps.lo[i] = lo[i]; taps.hi[i] = hi[i]; }
    g.nrun = (int)ceil_div(g.W, R);
#ifndef WV_SWT_PITCH_PAD
#define WV_SWT_PITCH_PAD 4
#endif
    g.P = g.nrun * R + WV_SWT_PITCH_PAD;          // multiple of 4; rows hold whole runs
    const size_t lds = (size_t)2 * 2 * TH * g.P * sizeof(float);  // 2 buffers x 2 planes
    // shapes this kernel does not take: the caller falls back to the tiled kernels
    if (lds > (size_t)kMaxLdsBytes - 2048 || TH * g.nrun > NT || g.W > NT) return 1;
    (void)HALO;
    if (g.W < R + HALO || g.H < TH + 2 * HALO || (uint64_t)g.H * g.W * 4 >= (1ull << 30)) return 1;
    auto kern = k_swt_slide<L, NLEV, R, TH, NT, MINW, InT, LAYOUT, BF16>;
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
    const char *env = ::wv::tune("WV_SWT_WG_PER_CU");
    const int per_cu = env && atoi(env) > 0 ? atoi(env) : std::min(by_lds, std::max(1, MINW * 256 / (2 * NT)));
    const int64_t planes = (int64_t)g.B * g.C;
    int64_t grid = std::min<int64_t>(planes, (int64_t)num_cu * per_cu);
    const char *cenv = ::wv::tune("WV_SWT_COAL");
    g.coal = std::is_same<InT, uint8_t>::value && LAYOUT == 0 && R == 16 && g.W % 16 == 0 && g.nrun <= 16 &&
             (int64_t)g.H * g.W % 16 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && !(cenv && atoi(cenv) == 0);
    const char *xenv = ::wv::tune("WV_SWT_XCD");
    g.nxcd = xenv ? std::max(1, atoi(xenv)) : 8;
    if (grid < planes) grid -= grid % g.nxcd;      // persistent launch: same number of workgroups on every XCD
    if (grid <= 0 || grid % g.nxcd) g.nxcd = 1, grid = std::min<int64_t>(planes, (int64_t)num_cu * per_cu);
    constexpr bool kHasStampBuild = (L == 4 && NLEV == 3) || (L == 2 && NLEV == 1);   // the two shipped configs
    if constexpr (kHasStampBuild)
    if (::wv::tune("WV_SWT_STAMPS")) {   // diagnostic build: run once, print where each role's cycles go
        auto kstamp = k_swt_slide<L, NLEV, R, TH, NT, MINW, InT, LAYOUT, BF16, true>;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kstamp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const size_t nw = (size_t)grid * (2 * NT / 64);
        unsigned long long *dbuf = nullptr;
        if (hipMalloc(&dbuf, nw * 8 * sizeof(unsigned long long)) != hipSuccess) WV_FAIL(WV_EHIP, "stamps: hipMalloc");
        g.stamps = dbuf;
        hipLaunchKernelGGL(kstamp, dim3((unsigned)grid), dim3(2 * NT), lds, st, (const InT *)in, out, g, taps);
        std::vector<unsigned long long> hbuf(nw * 8, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hbuf.data(), dbuf, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(dbuf);
        double a[2][6] = {{0}, {0}};
        size_t cnt[2] = {0, 0};
        for (size_t i = 0; i < nw; ++i) {
            const int role = (int)hbuf[8 * i + 3];
            for (int c = 0; c < 6; ++c) a[role][c] += (double)hbuf[8 * i + c];
            cnt[role]++;
        }
        fprintf(stderr, "[swt stamps] V waves=%zu: LDS read+lower(pl0) %.0f | rest(last+stores, pl1) %.0f | barrier %.0f\n", cnt[0],
                a[0][0] / cnt[0], a[0][1] / cnt[0], a[0][2] / cnt[0]);
        fprintf(stderr, "[swt stamps] H waves=%zu: wait loads+convert %.0f | issue next loads %.0f | lower levels %.0f | last level+LDS write %.0f | barrier %.0f\n",
                cnt[1], a[1][0] / cnt[1], a[1][4] / cnt[1], a[1][5] / cnt[1], a[1][1] / cnt[1], a[1][2] / cnt[1]);
        return WV_OK;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(2 * NT), lds, st, (const InT *)in, out, g, taps);
    WV_CHECK_LAUNCH("k_swt_slide");
    return WV_OK;
}

template <int L, int NLEV, int R, int TH, int NT, int MINW>
static int slide_types(const void *in, int in_dtype, void *out, const SlideGeom &g, const float *lo, const float *hi,
                       hipStream_t st)
{
    const bool nhwc3 = g.in_layout == WV_LAYOUT_NHWC && g.C == 3;
    if (g.in_layout != WV_LAYOUT_NCHW && !nhwc3) return 1;
#define WV_GO(T, LAY, BF) return launch_slide<L, NLEV, R, TH, NT, MINW, T, LAY, BF>(in, out, g, lo, hi, st)
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

// (taps, levels) with a sliding instantiation: the shipped configs (haar L1 = c0, db2 L3 = c1), the other levels of
// the 2- and 4-tap wavelets, and level 1 of the 8- and 10-tap wavelets the studies sweep (db4, bior4.4).  Everything
// else (and shapes outside the window below) runs on swt_fused.hip / swt.hip.
static bool slide_config(int L, int n)
{
    return ((L == 2 || L == 4) && n >= 1 && n <= 3) || ((L == 8 || L == 10) && n == 1);
}

bool swt_slide_covers(int L, int n, int W, int H)
{
    return slide_config(L, n) && (W % 4) == 0 && W <= 256 && W >= 40 && H >= 40;
}

int swt_slide_launch(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B, int C, int H,
                     int W, int n, const float *lo, const float *hi, int L, hipStream_t st, int out_layout,
                     int64_t band_stride)
{
    SlideGeom g{};
    g.B = B; g.C = C; g.H = H; g.W = W; g.in_layout = in_layout; g.out_bf16 = out_dtype == WV_DT_BF16;
    const size_t hw = (size_t)H * W;
    if (out_layout == WV_BANDS_OUTER) { g.pstride = hw; g.bstride = (size_t)band_stride; }
    else { g.pstride = 4 * hw; g.bstride = hw; }
#define WV_CFG(LL, NN) if (L == LL && n == NN) return slide_types<LL, NN, 16, 16, 256, 4>(in, in_dtype, out, g, lo, hi, st)
    WV_CFG(4, 3); WV_CFG(2, 1);
    WV_CFG(4, 1); WV_CFG(4, 2); WV_CFG(2, 2); WV_CFG(2, 3); WV_CFG(8, 1); WV_CFG(10, 1);
#undef WV_CFG
    return 1;
}

}  // namespace wv

// Shared host/device helpers for libwvhash (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/wvhash.h"

namespace wv {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// Value of a kernel-selection / tuning switch: always NULL in libwvhash.so (tune_release.cpp), the environment variable of
// that name in libwvhash_diag.so (tune_diag.cpp).
const char *tune(const char *name);

#define WV_FAIL(code, ...)          \
    do {                            \
        ::wv::set_error(__VA_ARGS__); \
        return (code);              \
    } while (0)

#define WV_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) WV_FAIL(WV_EINVAL, __VA_ARGS__); \
    } while (0)

// Checks the launch itself (asynchronous execution errors surface at the caller's next sync).
#define WV_CHECK_LAUNCH(what)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) WV_FAIL(WV_EHIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

// prepared-database blob = [distance-kernel tile image][ranking-kernel column image]
size_t dist_prepared_bytes(int64_t N, int words);
int dist_prepare(const uint64_t *db, void *dbP, int64_t N, int words, hipStream_t st);
size_t topk_prepared_bytes(int64_t N, int words);
int topk_prepare(const uint64_t *db, void *dbT, int64_t N, int words, hipStream_t st);

constexpr int kWave = 64;
constexpr int kMaxLdsBytes = 160 * 1024;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// count of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Wave64 inclusive scan on the DPP network (no LDS round trips): row_shr 1/2/4/8 inside each row of 16
// lanes, then row_bcast15 / row_bcast31 carry the row totals forward (the sequence LLVM's atomic
// optimizer emits for gfx9).  ~6 VALU instructions instead of 6 ds_bpermute round trips.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// sum over the wave, returned in every lane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

}  // namespace wv

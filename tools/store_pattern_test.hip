// Micro-test: how fast can the SWT output pattern be written?  (build: hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int H = 224, W = 224, TH = 32, TW = 112;

// A: thread = (plane, column): 32 rows x 2 bands of dword stores (the fused kernel's V pass)
__global__ __launch_bounds__(256) void kA(float *out, float v)
{
    const int tile = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int ty = tile / 2, tx = tile % 2, x0 = tx * TW, y0 = ty * TH;
    const size_t band = (size_t)H * W;
    for (int u = threadIdx.x; u < 2 * TW; u += 256) {
        const int pl = u / TW, x = u - pl * TW;
        const size_t base = (((size_t)b * 3 + c) * 4 + 2 * pl) * band + (size_t)y0 * W + x0 + x;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            out[base + (size_t)i * W] = v + i;
            out[base + band + (size_t)i * W] = v - i;
        }
    }
}
// B: thread = (band, row, 4 columns): float4 stores, lanes along x
__global__ __launch_bounds__(256) void kB(float *out, float v)
{
    const int tile = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int ty = tile / 2, tx = tile % 2, x0 = tx * TW, y0 = ty * TH;
    const size_t band = (size_t)H * W;
    for (int u = threadIdx.x; u < 4 * TH * (TW / 4); u += 256) {
        const int xq = u % (TW / 4), r = (u / (TW / 4)) % TH, bd = u / ((TW / 4) * TH);
        float4 val = make_float4(v, v + 1, v + 2, v + r);
        *reinterpret_cast<float4 *>(out + (((size_t)b * 3 + c) * 4 + bd) * band + (size_t)(y0 + r) * W + x0 + 4 * xq) = val;
    }
}
// C: flat streaming float4 fill
__global__ __launch_bounds__(256) void kC(float4 *out, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        out[i] = make_float4(v, v, v, v);
}
// D: full-width tiles (TH rows x 224): per band one contiguous TH*896 B block; thread = (plane, column), 512 threads
template <int THD, int NT>
__global__ __launch_bounds__(NT) void kD(float *out, float v)
{
    const int ty = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const int y0 = ty * THD;
    const size_t band = (size_t)H * W;
    for (int u = threadIdx.x; u < 2 * W; u += NT) {
        const int pl = u / W, x = u - pl * W;
        const size_t base = (((size_t)b * 3 + c) * 4 + 2 * pl) * band + (size_t)y0 * W + x;
#pragma unroll
        for (int i = 0; i < THD; ++i) {
            out[base + (size_t)i * W] = v + i;
            out[base + band + (size_t)i * W] = v - i;
        }
    }
}
// E: whole (image, channel) plane per workgroup, persistent over planes: 4 bands x 200 KB contiguous each
__global__ __launch_bounds__(256) void kE(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float4 *o = reinterpret_cast<float4 *>(out + (size_t)p * 4 * band);
        for (int i = threadIdx.x; i < 4 * H * W / 4; i += 256) o[i] = make_float4(v, v, v, v + i);
    }
}
// F: the sliding kernel's consumer pattern: persistent workgroups over planes, thread = column, per 16-row chunk
// 16 rows x 4 bands of dword stores (row-major, bands interleaved).  ORDER 1: band-major inside a chunk.
template <int ORDER, int NTH>
__global__ __launch_bounds__(NTH) void kF(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int t = threadIdx.x;
    if (t >= W) return;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float *o = out + (size_t)p * 4 * band + t;
        for (int y0 = 0; y0 < H; y0 += 16) {
            if (ORDER == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
#pragma unroll
                    for (int bd = 0; bd < 4; ++bd) o[bd * band + (size_t)(y0 + i) * W] = v + i;
            } else {
#pragma unroll
                for (int bd = 0; bd < 4; ++bd)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[bd * band + (size_t)(y0 + i) * W] = v + i;
            }
        }
    }
}
// G: wave = band: each wave of the workgroup streams one band of the plane (16-row chunks, dword stores)
__global__ __launch_bounds__(256) void kG(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int lane = threadIdx.x & 63, bd = threadIdx.x >> 6;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float *o = out + ((size_t)p * 4 + bd) * band;
        for (int y0 = 0; y0 < H; y0 += 16)
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (x * 64 + lane < W) o[(size_t)(y0 + i) * W + x * 64 + lane] = v + i;
    }
}
// Hh: wave = band, float4 stores: 16 rows x 896 B = 3584 float4... each lane 16 B, 56 lanes cover a row
__global__ __launch_bounds__(256) void kH(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int lane = threadIdx.x & 63, bd = threadIdx.x >> 6;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float4 *o = reinterpret_cast<float4 *>(out + ((size_t)p * 4 + bd) * band);
        for (int i = lane; i < H * W / 4; i += 64) o[i] = make_float4(v, v, v, v + i);
    }
}
// M: the sliding kernel's consumer with quad-transposed stores: thread = column, but each lane of a quad writes ONE
// row of the quad's 4 columns as a float4 (row y0 + 4g + (t & 3)), 4 bands, 16-row chunks
template <int NTH>
__global__ __launch_bounds__(NTH) void kM(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int t = threadIdx.x;
    if (t >= W) return;
    const int j = t & 3, c0 = t & ~3;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float *o = out + (size_t)p * 4 * band + c0;
        for (int y0 = 0; y0 < H; y0 += 16) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int bd = 0; bd < 4; ++bd)
                    *reinterpret_cast<float4 *>(o + bd * band + (size_t)(y0 + 4 * g + j) * W) = make_float4(v, v + g, v + bd, v + j);
        }
    }
}
// N: like M but a lane owns 4 consecutive rows' worth of ONE row segment of 16 columns?  -> thread = (row-in-group, 16-B column group):
// 64 lanes = 1 row x 56 float4 (one full 896-B row per wave instruction), waves take rows round-robin
__global__ __launch_bounds__(256) void kN(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane >= W / 4) return;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float *o = out + (size_t)p * 4 * band + 4 * lane;
        for (int y0 = 0; y0 < H; y0 += 16) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int bd = 0; bd < 4; ++bd)
                    *reinterpret_cast<float4 *>(o + bd * band + (size_t)(y0 + 4 * g + wv) * W) = make_float4(v, v + g, v + bd, v);
        }
    }
}

// Q: what torch's fill does: short-lived workgroups, one float4 per thread, consecutive workgroups = consecutive 4 KB
__global__ __launch_bounds__(256) void kQ(float4 *out, float v)
{
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = make_float4(v, v, v, v);
}
// R: the same with 4 float4 per thread (16 KB per workgroup), still short-lived
__global__ __launch_bounds__(256) void kR(float4 *out, float v)
{
    float4 *o = out + (size_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i * 256] = make_float4(v, v, v, v + i);
}
// T: persistent (2048 workgroups), but every workgroup writes long contiguous runs: 64 KB per iteration
__global__ __launch_bounds__(256) void kT(float4 *out, size_t n4, float v)
{
    for (size_t base = (size_t)blockIdx.x * 4096; base < n4; base += (size_t)gridDim.x * 4096)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[base + i * 256 + threadIdx.x] = make_float4(v, v, v, v + i);
}

// U: F's pattern into the BAND-MAJOR layout [4][planes][H][W] (what the models consume); CHUNK rows per burst
template <int CHUNK, int NTH>
__global__ __launch_bounds__(NTH) void kU(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W, bstride = (size_t)planes * band;
    const int t = threadIdx.x;
    if (t >= W) return;
    for (int p = blockIdx.x; p < planes; p += gridDim.x) {
        float *o = out + (size_t)p * band + t;
        for (int y0 = 0; y0 < H; y0 += CHUNK)
#pragma unroll
            for (int i = 0; i < CHUNK; ++i)
#pragma unroll
                for (int bd = 0; bd < 4; ++bd) o[bd * bstride + (size_t)(y0 + i) * W] = v + i;
    }
}
// V: two planes side by side per workgroup: 448 threads = 7 full waves (F wastes half of its fourth wave), reference layout
__global__ __launch_bounds__(448) void kV(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int t = threadIdx.x, pl = t / W, x = t - pl * W;
    for (int p = 2 * blockIdx.x; p < planes; p += 2 * gridDim.x) {
        float *o = out + (size_t)(p + pl) * 4 * band + x;
        for (int y0 = 0; y0 < H; y0 += 16)
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int bd = 0; bd < 4; ++bd) o[bd * band + (size_t)(y0 + i) * W] = v + i;
    }
}
// W: F with the planes dealt so that concurrently running workgroups write NEIGHBOURING planes (workgroup w takes planes
// w, w + G, ...: what F does) vs blocks of consecutive planes per workgroup (w * per .. w * per + per - 1)
__global__ __launch_bounds__(256) void kW(float *out, float v, int planes)
{
    const size_t band = (size_t)H * W;
    const int t = threadIdx.x;
    if (t >= W) return;
    const int per = (planes + gridDim.x - 1) / gridDim.x;
    for (int p = blockIdx.x * per; p < min(planes, (int)(blockIdx.x + 1) * per); ++p) {
        float *o = out + (size_t)p * 4 * band + t;
        for (int y0 = 0; y0 < H; y0 += 16)
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int bd = 0; bd < 4; ++bd) o[bd * band + (size_t)(y0 + i) * W] = v + i;
    }
}

int main()
{
    const int B = 2048;
    const size_t n = (size_t)B * 3 * 4 * H * W;
    float *out;
    CK(hipMalloc(&out, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 26; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(kA, dim3(14, 3, B), dim3(256), 0, 0, out, 1.0f);
            if (which == 1) hipLaunchKernelGGL(kB, dim3(14, 3, B), dim3(256), 0, 0, out, 1.0f);
            if (which == 2) hipLaunchKernelGGL(kC, dim3(256 * 8), dim3(256), 0, 0, (float4 *)out, n / 4, 1.0f);
            if (which == 3) hipLaunchKernelGGL((kD<32, 512>), dim3(7, 3, B), dim3(512), 0, 0, out, 1.0f);
            if (which == 4) hipLaunchKernelGGL((kD<16, 512>), dim3(14, 3, B), dim3(512), 0, 0, out, 1.0f);
            if (which == 5) hipLaunchKernelGGL((kD<56, 512>), dim3(4, 3, B), dim3(512), 0, 0, out, 1.0f);
            if (which == 6) hipLaunchKernelGGL(kE, dim3(256 * 4), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 7) hipLaunchKernelGGL((kF<0, 256>), dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 8) hipLaunchKernelGGL((kF<1, 256>), dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 9) hipLaunchKernelGGL((kF<0, 256>), dim3(1024), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 10) hipLaunchKernelGGL(kG, dim3(1024), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 11) hipLaunchKernelGGL(kH, dim3(1024), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 16) hipLaunchKernelGGL(kQ, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, (float4 *)out, 1.0f);
            if (which == 17) hipLaunchKernelGGL(kR, dim3((unsigned)(n / 4 / 1024)), dim3(256), 0, 0, (float4 *)out, 1.0f);
            if (which == 18) hipLaunchKernelGGL(kT, dim3(2048), dim3(256), 0, 0, (float4 *)out, n / 4, 1.0f);
            if (which == 19) hipLaunchKernelGGL((kU<16, 256>), dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 20) hipLaunchKernelGGL((kU<32, 256>), dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 21) hipLaunchKernelGGL(kV, dim3(256), dim3(448), 0, 0, out, 1.0f, 3 * B);
            if (which == 22) hipLaunchKernelGGL(kV, dim3(512), dim3(448), 0, 0, out, 1.0f, 3 * B);
            if (which == 23) hipLaunchKernelGGL(kW, dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 24) hipLaunchKernelGGL((kF<0, 256>), dim3(256), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 25) hipLaunchKernelGGL((kF<0, 256>), dim3(768), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 12) hipLaunchKernelGGL((kM<256>), dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 13) hipLaunchKernelGGL((kM<256>), dim3(1024), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 14) hipLaunchKernelGGL(kN, dim3(512), dim3(256), 0, 0, out, 1.0f, 3 * B);
            if (which == 15) hipLaunchKernelGGL(kN, dim3(1024), dim3(256), 0, 0, out, 1.0f, 3 * B);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("pattern %c: %.3f ms  %.0f GB/s\n", 'A' + which, best, n * 4 / best / 1e6);
    }
    return 0;
}

// Practical fp32 MFMA ceiling on this chip: bare v_mfma_f32_32x32x2_f32 loop, 3 independent accumulators per wave
// (the panel GEMM's inner loop without any operand traffic), one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        for (int iters : {200, 2000}) {
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)wgs * 4 * iters * 24 * 4096.0;
            printf("wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", wgs, iters, ms, fl / ms / 1e9);
        }
    }
    return 0;
}

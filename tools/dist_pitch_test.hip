// Micro-test: does the row pitch of the u8 distance matrix change the store throughput of column-tiled writes?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int Q = 2048, N = 25000;
__global__ __launch_bounds__(256) void kA(uint8_t *out, int ld, int qch, uint32_t v)
{
    const int n0 = blockIdx.x * 4096 + threadIdx.x * 16;
    if (n0 >= N) return;
    for (int qi = blockIdx.y * qch; qi < (blockIdx.y + 1) * qch; ++qi)
        *reinterpret_cast<uint4 *>(out + (size_t)qi * ld + n0) = make_uint4(v + qi, v, v, v);
}
__global__ __launch_bounds__(256) void kC(uint8_t *out, int ld, int rows, uint32_t v)
{
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = out + (size_t)(blockIdx.x * rows + r) * ld;
        for (int i = threadIdx.x * 16; i < N; i += 4096) *reinterpret_cast<uint4 *>(row + i) = make_uint4(v + r, v, v, v);
    }
}
__global__ __launch_bounds__(256) void kCnt(uint8_t *out, int ld, int rows, uint32_t v)
{
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = out + (size_t)(blockIdx.x * rows + r) * ld;
        for (int i = threadIdx.x * 16; i < N; i += 4096) {
            uint32_t *p = reinterpret_cast<uint32_t *>(row + i);
            __builtin_nontemporal_store(v + r, p); __builtin_nontemporal_store(v, p + 1);
            __builtin_nontemporal_store(v, p + 2); __builtin_nontemporal_store(v, p + 3);
        }
    }
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kAnt(uint8_t *out, int ld, int qch, uint32_t v)
{
    const int n0 = blockIdx.x * 4096 + threadIdx.x * 16;
    if (n0 >= N) return;
    for (int qi = blockIdx.y * qch; qi < (blockIdx.y + 1) * qch; ++qi) {
        u32x4 val = {v + (uint32_t)qi, v, v, v};
        __builtin_nontemporal_store(val, reinterpret_cast<u32x4 *>(out + (size_t)qi * ld + n0));
    }
}
int main()
{
    uint8_t *out;
    CK(hipMalloc(&out, (size_t)Q * 65536));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds[] = {25024};
    for (int ld : lds) {
        for (int which = 0; which < 5; ++which) {
            float best = 1e9f;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(kA, dim3(7, Q / 8), dim3(256), 0, 0, out, ld, 8, 1u);
                else if (which == 1) hipLaunchKernelGGL(kC, dim3(Q / 4), dim3(256), 0, 0, out, ld, 4, 1u);
                else if (which == 2) hipLaunchKernelGGL(kCnt, dim3(Q / 4), dim3(256), 0, 0, out, ld, 4, 1u);
                else if (which == 3) hipLaunchKernelGGL(kAnt, dim3(7, Q / 8), dim3(256), 0, 0, out, ld, 8, 1u);
                else hipLaunchKernelGGL(kAnt, dim3(7, Q / 2), dim3(256), 0, 0, out, ld, 2, 1u);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("ld %6d %s: %.1f us  %.0f GB/s\n", ld, which == 0 ? "A tiles" : which == 1 ? "C rows " : which == 2 ? "C rows nt" : which == 3 ? "A tiles nt q8" : "A tiles nt q2", best * 1e3, (double)Q * N / best / 1e6);
        }
    }
    return 0;
}

// Micro-test: store-side ceiling of the [Q=2048][N=25000 (pitch 25024)] u8 distance matrix (51 MB)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int Q = 2048, LD = 25024;
// A: current shape: WG = 4096-byte column tile x qch rows
__global__ __launch_bounds__(256) void kA(uint8_t *out, int qch, uint32_t v)
{
    const int n0 = blockIdx.x * 4096 + threadIdx.x * 16;
    if (n0 >= LD) return;
    for (int qi = blockIdx.y * qch; qi < (blockIdx.y + 1) * qch; ++qi)
        *reinterpret_cast<uint4 *>(out + (size_t)qi * LD + n0) = make_uint4(v + qi, v, v, v);
}
// B: flat streaming fill of the same bytes
__global__ __launch_bounds__(256) void kB(uint4 *out, size_t n16, uint32_t v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        out[i] = make_uint4(v, v, v, v + (uint32_t)i);
}
// C: WG owns full rows: rows_per_wg rows x whole pitch
__global__ __launch_bounds__(256) void kC(uint8_t *out, int rows, uint32_t v)
{
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = out + (size_t)(blockIdx.x * rows + r) * LD;
        for (int i = threadIdx.x * 16; i < LD; i += 4096) *reinterpret_cast<uint4 *>(row + i) = make_uint4(v + r, v, v, v);
    }
}
// E: 1024-thread WG owns `rows` full rows; thread t writes bytes [32t, 32t+32) of every row
__global__ __launch_bounds__(1024) void kE(uint8_t *out, int rows, uint32_t v)
{
    const int n0 = threadIdx.x * 32;
    if (n0 >= LD) return;
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = out + (size_t)(blockIdx.x * rows + r) * LD + n0;
        reinterpret_cast<uint4 *>(row)[0] = make_uint4(v + r, v, v, v);
        reinterpret_cast<uint4 *>(row)[1] = make_uint4(v, v + r, v, v);
    }
}
// F: WG owns `rows` full rows and walks them together in NT*16-byte steps (rows interleaved per step)
template <int NT>
__global__ __launch_bounds__(NT) void kF(uint8_t *out, int rows, uint32_t v)
{
    for (int i = threadIdx.x * 16; i < LD; i += NT * 16)
        for (int r = 0; r < rows; ++r)
            *reinterpret_cast<uint4 *>(out + (size_t)(blockIdx.x * rows + r) * LD + i) = make_uint4(v + r, v, v, v);
}
// D: WG = rows x segment of SEG bytes (SEG multiple of 16), grid (ceil(LD/SEG), Q/rows)
__global__ __launch_bounds__(256) void kD(uint8_t *out, int rows, int seg, uint32_t v)
{
    const int c0 = blockIdx.x * seg, c1 = min(c0 + seg, LD);
    for (int r = 0; r < rows; ++r) {
        uint8_t *row = out + (size_t)(blockIdx.y * rows + r) * LD;
        for (int i = c0 + threadIdx.x * 16; i < c1; i += 4096) *reinterpret_cast<uint4 *>(row + i) = make_uint4(v + r, v, v, v);
    }
}
int main()
{
    uint8_t *out;
    const size_t bytes = (size_t)Q * LD;
    CK(hipMalloc(&out, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"F 256thr 8 rows", "F 256thr 4 rows", "F 512thr 8 rows", "F 1024thr 8 rows", "F 256thr 16 rows", "E 1024thr 8 rows", "E 1024thr 4 rows", "D 8 rows x 12512", "D 8 rows x 8352", "D 4 rows x 12512", "D 16 rows x 12512", "D 8 rows x 6256", "A qch=8", "A qch=16", "A qch=4", "B flat 2048 WGs", "B flat 1024 WGs", "C 1 row/WG", "C 2 rows/WG", "C 4 rows/WG"};
    for (int which = -12; which < 8; ++which) {
        float best = 1e9f, avg = 0;
        const int reps = 20;
        for (int rep = 0; rep < reps + 2; ++rep) {
            CK(hipEventRecord(e0));
            switch (which) {
            case -12: hipLaunchKernelGGL(kF<256>, dim3(Q / 8), dim3(256), 0, 0, out, 8, 1u); break;
            case -11: hipLaunchKernelGGL(kF<256>, dim3(Q / 4), dim3(256), 0, 0, out, 4, 1u); break;
            case -10: hipLaunchKernelGGL(kF<512>, dim3(Q / 8), dim3(512), 0, 0, out, 8, 1u); break;
            case -9: hipLaunchKernelGGL(kF<1024>, dim3(Q / 8), dim3(1024), 0, 0, out, 8, 1u); break;
            case -8: hipLaunchKernelGGL(kF<256>, dim3(Q / 16), dim3(256), 0, 0, out, 16, 1u); break;
            case -7: hipLaunchKernelGGL(kE, dim3(Q / 8), dim3(1024), 0, 0, out, 8, 1u); break;
            case -6: hipLaunchKernelGGL(kE, dim3(Q / 4), dim3(1024), 0, 0, out, 4, 1u); break;
            case -5: hipLaunchKernelGGL(kD, dim3(2, Q / 8), dim3(256), 0, 0, out, 8, 12512, 1u); break;
            case -4: hipLaunchKernelGGL(kD, dim3(3, Q / 8), dim3(256), 0, 0, out, 8, 8352, 1u); break;
            case -3: hipLaunchKernelGGL(kD, dim3(2, Q / 4), dim3(256), 0, 0, out, 4, 12512, 1u); break;
            case -2: hipLaunchKernelGGL(kD, dim3(2, Q / 16), dim3(256), 0, 0, out, 16, 12512, 1u); break;
            case -1: hipLaunchKernelGGL(kD, dim3(4, Q / 8), dim3(256), 0, 0, out, 8, 6256, 1u); break;
            case 0: hipLaunchKernelGGL(kA, dim3(7, Q / 8), dim3(256), 0, 0, out, 8, 1u); break;
            case 1: hipLaunchKernelGGL(kA, dim3(7, Q / 16), dim3(256), 0, 0, out, 16, 1u); break;
            case 2: hipLaunchKernelGGL(kA, dim3(7, Q / 4), dim3(256), 0, 0, out, 4, 1u); break;
            case 3: hipLaunchKernelGGL(kB, dim3(2048), dim3(256), 0, 0, (uint4 *)out, bytes / 16, 1u); break;
            case 4: hipLaunchKernelGGL(kB, dim3(1024), dim3(256), 0, 0, (uint4 *)out, bytes / 16, 1u); break;
            case 5: hipLaunchKernelGGL(kC, dim3(Q), dim3(256), 0, 0, out, 1, 1u); break;
            case 6: hipLaunchKernelGGL(kC, dim3(Q / 2), dim3(256), 0, 0, out, 2, 1u); break;
            case 7: hipLaunchKernelGGL(kC, dim3(Q / 4), dim3(256), 0, 0, out, 4, 1u); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) { best = ms < best ? ms : best; avg += ms / reps; }
        }
        printf("%-18s best %.1f us avg %.1f us  %.0f GB/s (best)\n", names[which + 12], best * 1e3, avg * 1e3, bytes / best / 1e6);
    }
    return 0;
}

// How does v_mfma_f32_32x32x2_f32 round?  D = C + A[:,0] B[0,:] + A[:,1] B[1,:] per element: candidates
//   A: fma(a1, b1, fma(a0, b0, c))      B: fma(a0, b0, fma(a1, b1, c))
//   C: one rounding of the exact sum    D: c + (round(a0 b0) + round(a1 b1)) ...
// Random operands with mixed magnitudes; counts how many of the 1024 x trials outputs each candidate reproduces bit for bit.
// (host twin of wv_knn_float: csrc/host_knn.cpp follows whichever candidate matches every output)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k(const float *a, const float *b, const float *c, float *d)
{
    // lane l: row/col = l & 31, k = l >> 5
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;
        acc[e] = c[row * 32 + col];
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(lane & 31) * 2 + (lane >> 5)], b[(lane & 31) * 2 + (lane >> 5)], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;
        d[row * 32 + col] = acc[e];
    }
}

static float rnd(int spread)
{
    const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
    return ldexpf(m, rand() % (2 * spread + 1) - spread);
}

int main()
{
    float *da, *db, *dc, *dd;
    hipMalloc(&da, 64 * 4); hipMalloc(&db, 64 * 4); hipMalloc(&dc, 1024 * 4); hipMalloc(&dd, 1024 * 4);
    long nA = 0, nB = 0, nC = 0, nD = 0, n = 0;
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<float> a(64), b(64), c(1024), d(1024);
        const int spread = trial % 12;
        for (auto &x : a) x = rnd(spread);
        for (auto &x : b) x = rnd(spread);
        for (auto &x : c) x = rnd(spread + 2);
        hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
        for (int row = 0; row < 32; ++row)
            for (int col = 0; col < 32; ++col) {
                const float a0 = a[row * 2], a1 = a[row * 2 + 1], b0 = b[col * 2], b1 = b[col * 2 + 1], cc = c[row * 32 + col];
                const float A = fmaf(a1, b1, fmaf(a0, b0, cc)), B = fmaf(a0, b0, fmaf(a1, b1, cc));
                const float C = (float)((long double)a0 * b0 + (long double)a1 * b1 + (long double)cc);
                const float D = cc + (a0 * b0 + a1 * b1);
                const float got = d[row * 32 + col];
                nA += !memcmp(&got, &A, 4); nB += !memcmp(&got, &B, 4); nC += !memcmp(&got, &C, 4); nD += !memcmp(&got, &D, 4);
                ++n;
            }
    }
    printf("outputs %ld: A fma(a1,b1,fma(a0,b0,c)) %ld | B fma(a0,b0,fma(a1,b1,c)) %ld | C single rounding %ld | D c+(p0+p1) %ld\n", n, nA, nB,
           nC, nD);
    return 0;
}

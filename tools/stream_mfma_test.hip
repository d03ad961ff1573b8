// Feasibility probe for a one-workgroup-per-CU fused head kernel: every wave streams its OWN pre-packed B fragments
// (1 KiB per wave-instruction, lane-contiguous, shared by all workgroups -> L2 hits) straight into registers with a
// prefetch ring of D k-chunks, takes the A fragment from an LDS-resident 32 x 384 tile, and issues 12
// v_mfma_f32_32x32x2_f32 per chunk into three accumulators.  No barrier in the loop.  Question: how close to the bare
// MFMA loop (tools/mfma_peak_test.hip) does this get at 1 wave per SIMD, with 5.9 MB of weights per workgroup?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// MODE 0: as the fused head front runs it; 1: no refill loads (B stays in registers); 2: no LDS read (A stays in registers);
// 3: neither -- a bare MFMA loop with this loop's bookkeeping
template <int D, int MODE = 0>
__global__ __launch_bounds__(256, 1) void k_stream(const f32x4 *__restrict__ Wp, float *__restrict__ out, int nchunks,
                                                   size_t wave_stride4)
{
    constexpr int LDA = 388;
    __shared__ float At[32 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32 * LDA; i += 256) At[i] = (float)((i * 37 + blockIdx.x) & 255) * (1.0f / 256.0f) - 0.5f;
    __syncthreads();
    const f32x4 *wp = Wp + wv * wave_stride4 + lane;
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    f32x4 ring[D][3];
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
#pragma unroll
        for (int b = 0; b < 3; ++b) ring[d][b] = wp[(size_t)(d * 3 + b) * 64];
    const float *arow = At + r * LDA + 4 * h;
    // the stream is D chunks longer than what is consumed: every reload is unconditional, ring slots never move
    f32x4 av_n = *reinterpret_cast<const f32x4 *>(arow);
    const f32x4 *wq = wp + (size_t)D * 3 * 64;
    int ka = 0;
    for (int c = 0; c < nchunks; c += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // refill the slot the previous step consumed (stream position: D - 1 chunks ahead of this step)
            const int slot = (d + D - 1) % D;
            if (MODE == 0 || MODE == 2) {
#pragma unroll
                for (int b = 0; b < 3; ++b) ring[slot][b] = wq[(size_t)((d - 1) * 3 + b) * 64];
            }
            const f32x4 av = av_n;
            ka = ka + 8 == 384 ? 0 : ka + 8;
            if (MODE == 0 || MODE == 1) av_n = *reinterpret_cast<const f32x4 *>(arow + ka);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, ring[d][b].x, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, ring[d][b].y, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, ring[d][b].z, acc[b], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, ring[d][b].w, acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        wq += (size_t)D * 3 * 64;
    }
    float s = 0;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[b][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int D, int MODE = 0>
static void run(const f32x4 *Wp, float *out, int nchunks, size_t ws4, int wgs)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<D, MODE>), dim3(wgs), dim3(256), 0, 0, Wp, out, nchunks, ws4);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<D, MODE>), dim3(wgs), dim3(256), 0, 0, Wp, out, nchunks, ws4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double fl = (double)wgs * 4 * nchunks * 12 * 4096.0;
    printf("mode %d D=%d wgs=%d chunks/wave=%d: %.1f us  %.1f TFLOP/s  (weights per WG %.2f MB, L2->CU %.1f GB/s per CU)\n", MODE, D, wgs,
           nchunks, ms * 1e3, fl / ms / 1e9, 4.0 * nchunks * 3072 / 1e6, 4.0 * nchunks * 3072 / (ms * 1e-3) / 1e9 / 4);
}

int main()
{
    const int nchunks = 480;   // per wave: 5760 MFMAs, the whole fused head front
    const size_t ws4 = (size_t)nchunks * 3 * 64;
    std::vector<float> h((ws4 * 4 + 16 * 3 * 64) * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) * (1.0f / 1024.0f) - 0.5f;
    f32x4 *Wp;
    float *out;
    hipMalloc(&Wp, h.size() * 4);
    hipMemcpy(Wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 1024 * 256 * 4);
    for (int wgs : {256, 512}) {
        run<2>(Wp, out, nchunks, ws4, wgs);
        run<4>(Wp, out, nchunks, ws4, wgs);
        run<6>(Wp, out, nchunks, ws4, wgs);
        run<8>(Wp, out, nchunks, ws4, wgs);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(Wp, out, nchunks, ws4, 256);
        run<4, 1>(Wp, out, nchunks, ws4, 256);
        run<4, 2>(Wp, out, nchunks, ws4, 256);
        run<4, 3>(Wp, out, nchunks, ws4, 256);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    return 0;
}

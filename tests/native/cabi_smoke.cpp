// A consumer of the C ABI with no PyTorch and no Python in the process: plain hipMalloc'ed buffers, the entry
// points of include/wvhash.h, answers checked on the host against closed forms / brute force.
//   SWT   : haar level 1 of a uint8 batch against the 2x2 closed form
//   rank  : wv_pack_bits + wv_hamming_topk against a brute-force stable sort of popcount distances
//   mAP   : wv_map_at_k against the AP formula of accuracy_calculator.py:216-229 evaluated on the host
// Build: hipcc -O2 --offload-arch=gfx950 -I include -o tests/native/cabi_smoke tests/native/cabi_smoke.cpp \
//        -L image-retrieval-wavelet_amd/wvhash/_lib -lwvhash -Wl,-rpath,'$ORIGIN/../../image-retrieval-wavelet_amd/wvhash/_lib'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>
#include "wvhash.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define WVOK(x) do { int rc_ = (x); if (rc_ != WV_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, wv_last_error()); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main()
{
    if (wv_abi_version() != 5) { fprintf(stderr, "abi version %d\n", wv_abi_version()); return 1; }
    hipStream_t st;
    HIPOK(hipStreamCreate(&st));

    // ---------------------------------------------------------------- SWT haar level 1, uint8 [B,3,H,W]
    const int B = 3, C = 3, H = 48, W = 56;
    std::vector<uint8_t> img((size_t)B * C * H * W);
    for (auto &v : img) v = (uint8_t)(rnd() & 255);
    uint8_t *d_img; float *d_out; void *d_ws = nullptr;
    HIPOK(hipMalloc(&d_img, img.size()));
    HIPOK(hipMalloc(&d_out, img.size() * 4 * sizeof(float)));
    HIPOK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
    const float s = 0.70710678118654752440f, lo[2] = {s, s}, hi[2] = {-s, s};
    const size_t ws_bytes = wv_swt2d_workspace_bytes(B, C, H, W, 1, 2);
    if (ws_bytes) HIPOK(hipMalloc(&d_ws, ws_bytes));
    WVOK(wv_swt2d_forward(d_img, WV_DT_U8, WV_LAYOUT_NCHW, d_out, WV_DT_F32, B, C, H, W, 1, lo, hi, 2, d_ws, ws_bytes, st));
    HIPOK(hipStreamSynchronize(st));
    std::vector<float> out(img.size() * 4);
    HIPOK(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    double worst = 0;
    for (int p = 0; p < B * C; ++p)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                auto px = [&](int yy, int xx) { return (double)img[((size_t)p * H + (yy + H) % H) * W + (xx + W) % W] / 255.0; };
                // periodized haar, level 1: y[o] = f[0] x[o+1] + f[1] x[o]  (SURVEY.md 8 a-1 closed form)
                const double a = px(y, x), b = px(y, x + 1), c = px(y + 1, x), d = px(y + 1, x + 1);
                const double ref[4] = {(a + b + c + d) / 2, (a + b - c - d) / 2, (a - b + c - d) / 2, (a - b - c + d) / 2};
                for (int band = 0; band < 4; ++band) {
                    const double got = out[(((size_t)p * 4 + band) * H + y) * W + x];
                    worst = std::max(worst, std::fabs(got - ref[band]));
                }
            }
    if (worst > 2e-6) { fprintf(stderr, "SWT haar closed form: max |err| %.3g\n", worst); return 4; }

    // ---------------------------------------------------------------- pack + rank + AP
    const int Q = 7, N = 1000, nbits = 64, k = 200, LC = 20;
    std::vector<float> qc((size_t)Q * nbits), rc((size_t)N * nbits), ql((size_t)Q * LC), rl((size_t)N * LC);
    for (auto &v : qc) v = (rnd() & 1) ? 1.f : -1.f;
    for (auto &v : rc) v = (rnd() & 1) ? 1.f : -1.f;
    for (auto &v : ql) v = (rnd() % 5 == 0) ? 1.f : 0.f;
    for (auto &v : rl) v = (rnd() % 5 == 0) ? 1.f : 0.f;
    float *d_qc, *d_rc, *d_ql, *d_rl, *d_ap; uint64_t *d_qp, *d_rp, *d_qlp, *d_rlp; int32_t *d_idx, *d_nrel; uint8_t *d_dist;
    HIPOK(hipMalloc(&d_qc, qc.size() * 4)); HIPOK(hipMalloc(&d_rc, rc.size() * 4));
    HIPOK(hipMalloc(&d_ql, ql.size() * 4)); HIPOK(hipMalloc(&d_rl, rl.size() * 4));
    HIPOK(hipMalloc(&d_qp, Q * 8)); HIPOK(hipMalloc(&d_rp, N * 8)); HIPOK(hipMalloc(&d_qlp, Q * 8)); HIPOK(hipMalloc(&d_rlp, N * 8));
    HIPOK(hipMalloc(&d_idx, (size_t)Q * k * 4)); HIPOK(hipMalloc(&d_dist, (size_t)Q * k)); HIPOK(hipMalloc(&d_ap, Q * 4)); HIPOK(hipMalloc(&d_nrel, Q * 4));
    HIPOK(hipMemcpy(d_qc, qc.data(), qc.size() * 4, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_rc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_ql, ql.data(), ql.size() * 4, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(d_rl, rl.data(), rl.size() * 4, hipMemcpyHostToDevice));
    WVOK(wv_pack_bits(d_qc, nbits, d_qp, Q, nbits, 0, nullptr, st));
    WVOK(wv_pack_bits(d_rc, nbits, d_rp, N, nbits, 0, nullptr, st));
    WVOK(wv_pack_bits(d_ql, LC, d_qlp, Q, LC, 1, nullptr, st));
    WVOK(wv_pack_bits(d_rl, LC, d_rlp, N, LC, 1, nullptr, st));
    const size_t tk_bytes = wv_hamming_topk_workspace_bytes(Q, N, 1, k);
    void *d_tk; HIPOK(hipMalloc(&d_tk, tk_bytes));
    WVOK(wv_hamming_topk(d_qp, d_rp, d_idx, d_dist, Q, N, nbits, k, 0, d_tk, tk_bytes, st));
    WVOK(wv_map_at_k(d_idx, Q, k, d_qlp, d_rlp, 1, d_ap, d_nrel, st));
    HIPOK(hipStreamSynchronize(st));
    std::vector<int32_t> idx((size_t)Q * k), nrel(Q); std::vector<uint8_t> dist((size_t)Q * k); std::vector<float> ap(Q);
    HIPOK(hipMemcpy(idx.data(), d_idx, idx.size() * 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(dist.data(), d_dist, dist.size(), hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(ap.data(), d_ap, Q * 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(nrel.data(), d_nrel, Q * 4, hipMemcpyDeviceToHost));
    for (int qi = 0; qi < Q; ++qi) {
        std::vector<int> d(N), order(N);
        for (int n = 0; n < N; ++n) {
            int diff = 0;
            for (int b = 0; b < nbits; ++b) diff += qc[(size_t)qi * nbits + b] != rc[(size_t)n * nbits + b];
            d[n] = diff;
        }
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
        int hits = 0; double acc = 0;
        for (int p = 0; p < k; ++p) {
            if (idx[(size_t)qi * k + p] != order[p] || dist[(size_t)qi * k + p] != d[order[p]]) {
                fprintf(stderr, "rank mismatch q=%d p=%d: got (%d,%d) want (%d,%d)\n", qi, p, idx[(size_t)qi * k + p],
                        dist[(size_t)qi * k + p], order[p], d[order[p]]);
                return 5;
            }
            bool rel = false;
            for (int c = 0; c < LC; ++c) rel |= ql[(size_t)qi * LC + c] > 0 && rl[(size_t)order[p] * LC + c] > 0;
            if (rel) { ++hits; acc += (double)((float)hits / (float)(p + 1)); }
        }
        const float want = hits ? (float)(acc / hits) : 0.f;
        if (nrel[qi] != hits || std::fabs(ap[qi] - want) > 1e-6f) {
            fprintf(stderr, "AP mismatch q=%d: got %g (%d hits) want %g (%d hits)\n", qi, ap[qi], nrel[qi], want, hits);
            return 6;
        }
    }
    // error path: the library reports, it does not crash
    if (wv_hamming_topk(d_qp, d_rp, d_idx, d_dist, Q, N, nbits, N + 1, 0, d_tk, tk_bytes, st) != WV_EINVAL) return 7;
    printf("cabi_smoke ok: swt haar closed form err %.2g, %d ranked lists exact, AP exact; last error text: \"%s\"\n", worst, Q,
           wv_last_error());
    return 0;
}

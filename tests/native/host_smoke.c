/* Plain C consumer of the HOST side of the C ABI (gcc, no HIP header, no GPU): include/wvhash.h is valid C, and the `_cpu`
 * twins -- what CustomCalculator(device='cpu') and the DataLoader workers call -- work from a program that never touches a
 * GPU (BASELINE config c0: "CPU ... plumbing").  Checks tiny closed-form cases: Haar level-1 SWT of a constant image (cA = 2 c,
 * details 0 up to rounding: SURVEY 8(c) ii), packing, Hamming distances, the stable ranking and the AP formula of
 * accuracy_calculator.py:203-231.  Built by tests/native/Makefile, run by tests/test_abi.py on the CPU box. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "wvhash.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "host_smoke: %s failed (line %d): %s\n", #cond, __LINE__, wv_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void)
{
    CHECK(wv_abi_version() == 5);
    /* ---- SWT twin: constant uint8 image 200 -> cA = 2 * 200/255, details exactly 0 */
    enum { H = 8, W = 8 };
    uint8_t img[3 * H * W];
    float out[3 * 4 * H * W];
    const float s = 0.70710678118654752440f, lo[2] = {s, s}, hi[2] = {-s, s};
    memset(img, 200, sizeof(img));
    CHECK(wv_swt2d_forward_cpu(img, WV_DT_U8, WV_LAYOUT_NCHW, out, 1, 3, H, W, 1, lo, hi, 2) == WV_OK);
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < H * W; ++i) {
            CHECK(fabsf(out[(c * 4 + 0) * H * W + i] - 2.0f * 200.0f / 255.0f) < 1e-6f);
            for (int b = 1; b < 4; ++b) CHECK(fabsf(out[(c * 4 + b) * H * W + i]) < 1e-7f);   /* (fused multiply-adds: not exactly 0) */
        }
    CHECK(wv_swt2d_forward_cpu(img, WV_DT_U8, WV_LAYOUT_NCHW, out, 1, 3, 7, 8, 1, lo, hi, 2) != WV_OK);   /* 7 rows: not a multiple of 2 */
    /* ---- ranking twins: 4 database codes of 8 bits, 2 queries */
    const float db[4][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1, -1, -1}, {1, 1, 1, 1, 1, 1, 1, -1}};
    const float q[2][8] = {{1, 1, 1, 1, 1, 1, 1, 1}, {-1, -1, -1, -1, -1, -1, -1, 1}};
    uint64_t dbp[4], qp[2];
    int32_t bad = 0;
    CHECK(wv_pack_bits_cpu(&db[0][0], 8, dbp, 4, 8, 0, &bad) == WV_OK && bad == 0);
    CHECK(wv_pack_bits_cpu(&q[0][0], 8, qp, 2, 8, 0, &bad) == WV_OK && bad == 0);
    CHECK(dbp[0] == 0xff && dbp[1] == 0x0f && dbp[2] == 0x00 && dbp[3] == 0x7f);
    uint8_t dist[2][4];
    CHECK(wv_hamming_dist_cpu(qp, dbp, &dist[0][0], 4, 2, 4, 1) == WV_OK);
    CHECK(dist[0][0] == 0 && dist[0][1] == 4 && dist[0][2] == 8 && dist[0][3] == 1);
    CHECK(dist[1][0] == 7 && dist[1][1] == 5 && dist[1][2] == 1 && dist[1][3] == 8);
    int32_t idx[2][3];
    uint8_t dk[2][3];
    CHECK(wv_hamming_topk_cpu(qp, dbp, &idx[0][0], &dk[0][0], 2, 4, 8, 3, 100) == WV_OK);
    CHECK(idx[0][0] == 100 && idx[0][1] == 103 && idx[0][2] == 101 && dk[0][2] == 4);
    CHECK(idx[1][0] == 102 && idx[1][1] == 101 && idx[1][2] == 100 && dk[1][0] == 1);
    /* AP: query 0 shares a label with rows 0 and 1 (ranks 1 and 3): AP = (1/1 + 2/3) / 2; query 1 with nothing: 0 */
    const uint64_t qlab[2] = {0x1, 0x4}, dblab[4] = {0x1, 0x3, 0x2, 0x2};
    int32_t lists[2][3];
    for (int i = 0; i < 6; ++i) (&lists[0][0])[i] = (&idx[0][0])[i] - 100;
    float ap[2];
    int32_t nrel[2];
    CHECK(wv_map_at_k_cpu(&lists[0][0], 3, 2, 3, qlab, dblab, 1, ap, nrel) == WV_OK);
    CHECK(nrel[0] == 2 && nrel[1] == 0 && ap[1] == 0.0f && fabsf(ap[0] - (1.0f + 2.0f / 3.0f) / 2.0f) < 1e-7f);
    uint32_t hits[2][3], counts[8];
    CHECK(wv_hit_prefix_cpu(&lists[0][0], 2, 3, qlab, dblab, 1, &hits[0][0]) == WV_OK);
    CHECK(hits[0][0] == 1 && hits[0][1] == 1 && hits[0][2] == 2 && hits[1][2] == 0);
    CHECK(wv_bit_counts_cpu(dbp, 4, 8, counts) == WV_OK && counts[0] == 3 && counts[7] == 1);
    /* ---- real-valued k-NN twin on the same +-1 rows: inner products 8 - 2 * hamming, squared L2 = 4 * hamming; ties (none
     * here) by ascending row */
    int32_t ki[2][3];
    float kv[2][3];
    CHECK(wv_knn_float_cpu(&q[0][0], &db[0][0], 2, 4, 8, WV_METRIC_IP, 3, &ki[0][0], &kv[0][0]) == WV_OK);
    CHECK(ki[0][0] == 0 && ki[0][1] == 3 && ki[0][2] == 1 && kv[0][0] == 8.0f && kv[0][1] == 6.0f && kv[0][2] == 0.0f);
    CHECK(wv_knn_float_cpu(&q[0][0], &db[0][0], 2, 4, 8, WV_METRIC_L2_SQUARED, 3, &ki[0][0], &kv[0][0]) == WV_OK);
    CHECK(ki[1][0] == 2 && ki[1][1] == 1 && ki[1][2] == 0 && kv[1][0] == 4.0f && kv[1][1] == 20.0f && kv[1][2] == 28.0f);
    CHECK(wv_knn_float_cpu(&q[0][0], &db[0][0], 2, 4, 8, WV_METRIC_L2, 1, &ki[0][0], &kv[0][0]) == WV_OK && kv[0][1] == 2.0f);
    /* ---- hashing tail twin: identity-like hash matrix on the first database row pair, no BatchNorm */
    {
        float hw[2][8] = {{1, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, -1}}, logits[2][2], codes[2][2];
        uint64_t pk[2];
        CHECK(wv_hash_tail_cpu(&db[0][0], 2, 8, &hw[0][0], NULL, NULL, NULL, NULL, NULL, 0.0f, 2, &logits[0][0], &codes[0][0], pk) == WV_OK);
        CHECK(logits[0][0] == 1.0f && logits[0][1] == -1.0f && logits[1][1] == 1.0f && codes[1][1] == 1.0f && pk[0] == 1 && pk[1] == 3);
    }
    /* argument validation answers with a code and a message, never a crash */
    CHECK(wv_hamming_topk_cpu(qp, dbp, &idx[0][0], NULL, 2, 4, 8, 5, 0) == WV_EINVAL && strstr(wv_last_error(), "k=5"));
    CHECK(wv_knn_float_cpu(&q[0][0], &db[0][0], 2, 4, 8, WV_METRIC_IP, 5, &ki[0][0], &kv[0][0]) == WV_EINVAL);   /* k > N */
    printf("host_smoke ok\n");
    return 0;
}

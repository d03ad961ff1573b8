// Host twin of wv_knn_float ("wv_l2_topk / wv_ip_topk ... _cpu twins of each taking host pointers"):
// get_knn_torch's cosine / l2 branches (/root/reference/main/engine/get_knn.py:60-71) and the faiss IndexFlatL2 flavour
// (:38-39,55) on HOST pointers, for CustomCalculator(device='cpu') with real-valued embeddings.
//
// Same results as the gfx950 path, bit for bit:
//  * v_mfma_f32_32x32x2_f32 IS an fmaf chain -- D = fma(a1, b1, fma(a0, b0, C)), k = 0 before k = 1, every one of 204,800
//    random outputs (tools/mfma_order_test.hip) -- and k_scores* feeds it k = 8c + e (lane half 0) and 8c + 4 + e (half 1)
//    for e = 0..3 of every chunk c of eight; the zero padding beyond D adds exact zeros.  The dot product below walks k in
//    that order with fused multiply-adds (the FMA instruction when the CPU has it, fmaf otherwise: the same bits);
//  * |x|^2 as k_row_sqnorm forms it: lane l of a wave sums k = l, l + 64, ... with fmaf, the 64 partial sums meet in the
//    xor butterfly 32, 16, ..., 1;  d2 = max(0, fma(-2, q.r, |q|^2 + |r|^2)), ranked squared, the root taken of the k results;
//  * ranking: ascending (order-preserving key of the score, database row) -- what both GPU rankings produce.
// No HIP call, no thread, no global state.  Product code: shares nothing with oracle/.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/wvhash.h"

namespace wv {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
}

#define HK_FAIL(code, ...)            \
    do {                              \
        ::wv::set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)
#define HK_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) HK_FAIL(WV_EINVAL, __VA_ARGS__); \
    } while (0)

namespace {

constexpr int kLanes = 8;     // database rows per vector
constexpr int kBlocks = 8;    // vectors in flight per query: eight independent fma chains

inline uint32_t float_to_key(float v, bool descending)
{
    v += 0.0f;  // -0 -> +0
    uint32_t u;
    memcpy(&u, &v, 4);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return descending ? ~u : u;
}
inline float key_to_float(uint32_t u, bool descending)
{
    if (descending) u = ~u;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    float v;
    memcpy(&v, &u, 4);
    return v;
}

// k in the order the matrix cores accumulate it
std::vector<int> accumulation_order(int D)
{
    std::vector<int> ord;
    ord.reserve(D);
    for (int c = 0; 8 * c < D; ++c)
        for (int e = 0; e < 4; ++e) {
            if (8 * c + e < D) ord.push_back(8 * c + e);
            if (8 * c + 4 + e < D) ord.push_back(8 * c + 4 + e);
        }
    return ord;
}

float sqnorm_like_the_kernel(const float *x, int D)
{
    float s[64];
    for (int l = 0; l < 64; ++l) {
        float a = 0.f;
        for (int k = l; k < D; k += 64) a = fmaf(x[k], x[k], a);
        s[l] = a;
    }
    for (int d = 32; d > 0; d >>= 1) {
        float t[64];
        for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ d];
        memcpy(s, t, sizeof(s));
    }
    return s[0];
}

// scores of one query against kBlocks * kLanes database rows of the interleaved image bt[block][j][lane]
void dots_plain(const float *qo, const float *bt, int D, float *out)
{
    for (int b = 0; b < kBlocks; ++b)
        for (int l = 0; l < kLanes; ++l) {
            float acc = 0.f;
            const float *col = bt + (size_t)b * D * kLanes + l;
            for (int j = 0; j < D; ++j) acc = fmaf(qo[j], col[(size_t)j * kLanes], acc);
            out[b * kLanes + l] = acc;
        }
}
#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) void dots_fma(const float *qo, const float *bt, int D, float *out)
{
    __m256 acc[kBlocks];
    for (int b = 0; b < kBlocks; ++b) acc[b] = _mm256_setzero_ps();
    const size_t pitch = (size_t)D * kLanes;
    for (int j = 0; j < D; ++j) {
        const __m256 a = _mm256_broadcast_ss(qo + j);
        const float *p = bt + (size_t)j * kLanes;
        for (int b = 0; b < kBlocks; ++b) acc[b] = _mm256_fmadd_ps(a, _mm256_loadu_ps(p + b * pitch), acc[b]);
    }
    for (int b = 0; b < kBlocks; ++b) _mm256_storeu_ps(out + b * kLanes, acc[b]);
}
#endif
void dots(const float *qo, const float *bt, int D, float *out)
{
#if defined(__x86_64__)
    static const int has = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    if (has) return dots_fma(qo, bt, D, out);
#endif
    dots_plain(qo, bt, D, out);
}

}  // namespace

extern "C" int wv_knn_float_cpu(const float *q, const float *db, int Q, int64_t N, int D, int metric, int k, int32_t *idx,
                                float *val)
{
    HK_REQUIRE(q && db && idx && val, "knn_float_cpu: null buffer");
    HK_REQUIRE(Q >= 0 && N >= 1 && D >= 1, "knn_float_cpu: bad shape Q=%d N=%lld D=%d", Q, (long long)N, D);
    HK_REQUIRE(metric == WV_METRIC_IP || metric == WV_METRIC_L2 || metric == WV_METRIC_L2_SQUARED, "knn_float_cpu: metric %d",
               metric);
    HK_REQUIRE(k >= 1 && k <= N, "knn_float_cpu: k=%d must be in [1, N=%lld] (torch.topk raises too)", k, (long long)N);
    HK_REQUIRE(N <= (1ll << 26), "knn_float_cpu: N=%lld above the supported 2^26 rows", (long long)N);
    if (Q == 0) return WV_OK;
    const bool l2 = metric != WV_METRIC_IP, desc = !l2;
    const std::vector<int> ord = accumulation_order(D);
    // database rows interleaved eight by eight, k in accumulation order (rows beyond N are zeros and never ranked)
    const int64_t group = (int64_t)kBlocks * kLanes, Np = (N + group - 1) / group * group;
    std::vector<float> bt((size_t)Np * D, 0.f);
    for (int64_t n = 0; n < N; ++n) {
        float *dst = bt.data() + (size_t)(n / kLanes) * D * kLanes + n % kLanes;
        const float *src = db + (size_t)n * D;
        for (int j = 0; j < D; ++j) dst[(size_t)j * kLanes] = src[ord[j]];
    }
    std::vector<float> dbn;
    if (l2) {
        dbn.resize((size_t)N);
        for (int64_t n = 0; n < N; ++n) dbn[(size_t)n] = sqnorm_like_the_kernel(db + (size_t)n * D, D);
    }
    std::vector<float> qo((size_t)D), row((size_t)Np);
    std::vector<uint64_t> keyed((size_t)N);
    for (int qi = 0; qi < Q; ++qi) {
        const float *qr = q + (size_t)qi * D;
        for (int j = 0; j < D; ++j) qo[(size_t)j] = qr[ord[j]];
        for (int64_t n0 = 0; n0 < Np; n0 += group) dots(qo.data(), bt.data() + (size_t)n0 * D, D, row.data() + n0);
        const float qn = l2 ? sqnorm_like_the_kernel(qr, D) : 0.f;
        for (int64_t n = 0; n < N; ++n) {
            float v = row[(size_t)n];
            if (l2) v = fmaxf(0.f, fmaf(-2.f, v, qn + dbn[(size_t)n]));
            keyed[(size_t)n] = ((uint64_t)float_to_key(v, desc) << 32) | (uint32_t)n;
        }
        if ((int64_t)k < N) std::nth_element(keyed.begin(), keyed.begin() + k, keyed.end());
        std::sort(keyed.begin(), keyed.begin() + k);
        for (int j = 0; j < k; ++j) {
            const uint64_t e = keyed[(size_t)j];
            const float v = key_to_float((uint32_t)(e >> 32), desc);
            idx[(size_t)qi * k + j] = (int32_t)(uint32_t)e;
            val[(size_t)qi * k + j] = metric == WV_METRIC_L2 ? sqrtf(v) : v;
        }
    }
    return WV_OK;
}

extern "C" int wv_rank_scores_cpu(const float *S, int Q, int64_t N, int k, int flags, int32_t *idx, float *val)
{
    HK_REQUIRE(S && idx && val, "rank_scores_cpu: null buffer");
    HK_REQUIRE(Q >= 0 && N >= 1, "rank_scores_cpu: bad shape Q=%d N=%lld", Q, (long long)N);
    HK_REQUIRE(k >= 1 && k <= N, "rank_scores_cpu: k=%d must be in [1, N=%lld]", k, (long long)N);
    HK_REQUIRE(N <= (1ll << 26), "rank_scores_cpu: N=%lld above the supported 2^26 columns", (long long)N);
    HK_REQUIRE((flags & ~(WV_RANK_DESCENDING | WV_RANK_SQRT)) == 0, "rank_scores_cpu: flags %d", flags);
    const bool desc = (flags & WV_RANK_DESCENDING) != 0, root = (flags & WV_RANK_SQRT) != 0;
    std::vector<uint64_t> keyed((size_t)N);
    for (int qi = 0; qi < Q; ++qi) {
        const float *row = S + (size_t)qi * N;
        for (int64_t n = 0; n < N; ++n) keyed[(size_t)n] = ((uint64_t)float_to_key(row[n], desc) << 32) | (uint32_t)n;
        if ((int64_t)k < N) std::nth_element(keyed.begin(), keyed.begin() + k, keyed.end());
        std::sort(keyed.begin(), keyed.begin() + k);
        for (int j = 0; j < k; ++j) {
            const uint64_t e = keyed[(size_t)j];
            const float v = key_to_float((uint32_t)(e >> 32), desc);
            idx[(size_t)qi * k + j] = (int32_t)(uint32_t)e;
            val[(size_t)qi * k + j] = root ? sqrtf(v) : v;
        }
    }
    return WV_OK;
}

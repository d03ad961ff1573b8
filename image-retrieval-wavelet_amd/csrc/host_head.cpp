// Host twins of the head entry points ("wv_band_attn_pool ... _cpu twins of each taking host pointers"):
//   wv_band_attn_pool_cpu <- wv_band_attn_pool   CrossAttentionBottleneckHead*.forward in eval mode
//                                                (/root/reference/main/models/multi_dino_attention.py:1111-1141; :1030, :568, :448)
//   wv_hash_tail_cpu      <- wv_hash_tail        hash_fc -> BatchNorm1d(eval) -> sign (+ bit packing)  (:829-833)
// for a model whose tensors live on the host (module.to('cpu') -- explicit, never a silent fallback).
//
// fp32 throughout, like the kernels; every contraction is the same on any machine: eight interleaved fused-multiply-add
// chains (k mod 8) combined in one fixed tree -- the FMA instruction when the CPU has it, fmaf otherwise, the same bits.
// Against the GPU path the outputs agree to fp32 rounding (the kernels sum in MFMA order and use an erfc-polynomial GELU;
// tests hold both to the reference-made golden vectors with the same tolerance, 5e-5), the codes and packed words wherever
// |logit| exceeds that.  No HIP call, no thread, no global state.  Product code: shares nothing with oracle/.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/wvhash.h"

namespace wv {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
}

#define HH_FAIL(code, ...)            \
    do {                              \
        ::wv::set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)
#define HH_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) HH_FAIL(WV_EINVAL, __VA_ARGS__); \
    } while (0)

namespace {

constexpr int MB = 4;   // rows of x per pass over a weight row

inline float tree8(const float *a) { return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])); }

// y[m][n] = bias[n] + sum_k x[m][k] w[n][k]   (x: M rows of pitch ldx, w: N rows of K, K % 8 == 0; y pitch ldy)
void linear_plain(const float *x, int64_t ldx, int M, const float *w, const float *bias, int N, int K, float *y, int64_t ldy)
{
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float *xr = x + m * ldx, *wr = w + (int64_t)n * K;
            for (int k = 0; k < K; k += 8)
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xr[k + j], wr[k + j], acc[j]);
            y[m * ldy + n] = tree8(acc) + (bias ? bias[n] : 0.f);
        }
}
#if defined(__x86_64__)
__attribute__((target("avx2,fma"))) void linear_fma(const float *x, int64_t ldx, int M, const float *w, const float *bias, int N,
                                                    int K, float *y, int64_t ldy)
{
    for (int m0 = 0; m0 < M; m0 += MB) {
        const int mb = M - m0 < MB ? M - m0 : MB;
        for (int n = 0; n < N; ++n) {
            __m256 acc[MB];
            for (int i = 0; i < MB; ++i) acc[i] = _mm256_setzero_ps();
            const float *wr = w + (int64_t)n * K;
            for (int k = 0; k < K; k += 8) {
                const __m256 wv = _mm256_loadu_ps(wr + k);
                for (int i = 0; i < mb; ++i) acc[i] = _mm256_fmadd_ps(_mm256_loadu_ps(x + (m0 + i) * ldx + k), wv, acc[i]);
            }
            for (int i = 0; i < mb; ++i) {
                float a[8];
                _mm256_storeu_ps(a, acc[i]);
                y[(m0 + i) * ldy + n] = tree8(a) + (bias ? bias[n] : 0.f);
            }
        }
    }
}
#endif
void linear(const float *x, int64_t ldx, int M, const float *w, const float *bias, int N, int K, float *y, int64_t ldy)
{
#if defined(__x86_64__)
    static const int has = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
    if (has) return linear_fma(x, ldx, M, w, bias, N, K, y, ldy);
#endif
    linear_plain(x, ldx, M, w, bias, N, K, y, ldy);
}

void layer_norm(float *x, int E, const float *w, const float *b, float eps)
{
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < E; ++i) mean += x[i];
    mean /= E;
    for (int i = 0; i < E; ++i) var += ((double)x[i] - mean) * ((double)x[i] - mean);
    const float inv = (float)(1.0 / sqrt(var / E + (double)eps)), mu = (float)mean;
    for (int i = 0; i < E; ++i) x[i] = (x[i] - mu) * inv * w[i] + b[i];
}

inline float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

}  // namespace

extern "C" int wv_band_attn_pool_cpu(const wv_head_params *p, const float *feats, int B, float *out)
{
    HH_REQUIRE(p && feats && out, "band_attn_pool_cpu: null buffer");
    const int E = p->embed_dim, H = p->num_heads, Nq = p->num_queries, S = p->num_tokens;
    HH_REQUIRE(E >= 8 && E % 8 == 0 && H >= 1 && E % H == 0 && Nq >= 1 && S >= 1 && S <= 64 && B >= 0,
               "band_attn_pool_cpu: bad configuration E=%d heads=%d Nq=%d S=%d B=%d", E, H, Nq, S, B);
    HH_REQUIRE(p->q_eff && p->in_proj_w && p->in_proj_b && p->attn_out_w && p->attn_out_b && p->norm1_w && p->norm1_b &&
                   p->mlp0_w && p->mlp0_b && p->mlp2_w && p->mlp2_b && p->out_w && p->out_b && p->norm2_w && p->norm2_b,
               "band_attn_pool_cpu: null parameter");
    if (B == 0) return WV_OK;
    const int d = E / H;
    const float scale = 1.f / sqrtf((float)d);
    // Q = q_eff Wq^T + bq: the query tokens are parameters, projected once (p->q_proj is a DEVICE pointer in the GPU path;
    // the host twin always projects)
    std::vector<float> Qp((size_t)Nq * E);
    linear(p->q_eff, E, Nq, p->in_proj_w, p->in_proj_b, E, E, Qp.data(), E);
    std::vector<float> kv((size_t)S * E), KV((size_t)S * 2 * E), ctx((size_t)Nq * E), x((size_t)Nq * E), hid((size_t)Nq * 4 * E),
        y((size_t)Nq * E), pooled((size_t)E);
    const int64_t sB = (int64_t)B * E;
    for (int b = 0; b < B; ++b) {
        for (int s = 0; s < S; ++s) memcpy(kv.data() + (size_t)s * E, feats + s * sB + (int64_t)b * E, sizeof(float) * E);
        // K | V = kv [Wk; Wv]^T + [bk; bv]  (rows E..3E of the packed in-projection)
        linear(kv.data(), E, S, p->in_proj_w + (size_t)E * E, p->in_proj_b + E, 2 * E, E, KV.data(), 2 * E);
        for (int h = 0; h < H; ++h)
            for (int nq = 0; nq < Nq; ++nq) {
                float sc[64], mx = -INFINITY;
                const float *qh = Qp.data() + (size_t)nq * E + h * d;
                for (int s = 0; s < S; ++s) {
                    const float *kh = KV.data() + (size_t)s * 2 * E + h * d;
                    float a = 0.f;
                    for (int j = 0; j < d; ++j) a = fmaf(qh[j], kh[j], a);
                    sc[s] = a * scale;
                    mx = fmaxf(mx, sc[s]);
                }
                float den = 0.f;
                for (int s = 0; s < S; ++s) { sc[s] = expf(sc[s] - mx); den += sc[s]; }
                float *o = ctx.data() + (size_t)nq * E + h * d;
                for (int j = 0; j < d; ++j) {
                    float a = 0.f;
                    for (int s = 0; s < S; ++s) a = fmaf(sc[s] / den, KV[(size_t)s * 2 * E + E + h * d + j], a);
                    o[j] = a;
                }
            }
        // x = LN1(q + attn.out_proj(ctx));  x = x + mlp.2(GELU(mlp.0(x)))
        linear(ctx.data(), E, Nq, p->attn_out_w, p->attn_out_b, E, E, x.data(), E);
        for (int nq = 0; nq < Nq; ++nq) {
            float *xr = x.data() + (size_t)nq * E;
            for (int i = 0; i < E; ++i) xr[i] += p->q_eff[(size_t)nq * E + i];
            layer_norm(xr, E, p->norm1_w, p->norm1_b, p->ln_eps);
        }
        linear(x.data(), E, Nq, p->mlp0_w, p->mlp0_b, 4 * E, E, hid.data(), 4 * E);
        for (size_t i = 0; i < hid.size(); ++i) hid[i] = gelu_erf(hid[i]);
        linear(hid.data(), 4 * E, Nq, p->mlp2_w, p->mlp2_b, E, 4 * E, y.data(), E);
        for (size_t i = 0; i < x.size(); ++i) x[i] += y[i];
        // read-out: mean over the queries then Linear(E -> E), or Linear(Nq E -> E) on the concatenation; LN2
        float *ob = out + (size_t)b * E;
        if (p->pool_mean) {
            for (int i = 0; i < E; ++i) {
                float a = 0.f;
                for (int nq = 0; nq < Nq; ++nq) a += x[(size_t)nq * E + i];
                pooled[(size_t)i] = a / (float)Nq;
            }
            linear(pooled.data(), E, 1, p->out_w, p->out_b, E, E, ob, E);
        } else {
            linear(x.data(), (int64_t)Nq * E, 1, p->out_w, p->out_b, E, Nq * E, ob, E);
        }
        layer_norm(ob, E, p->norm2_w, p->norm2_b, p->ln_eps);
    }
    return WV_OK;
}

extern "C" int wv_hash_tail_cpu(const float *fused, int B, int E, const float *hash_w, const float *hash_b, const float *bn_w,
                                const float *bn_b, const float *bn_mean, const float *bn_var, float bn_eps, int nbits,
                                float *logits_out, float *codes_out, uint64_t *packed_out)
{
    HH_REQUIRE(fused && hash_w, "hash_tail_cpu: null buffer");
    HH_REQUIRE(B >= 0 && E >= 8 && E % 8 == 0 && nbits >= 1, "hash_tail_cpu: bad shape B=%d E=%d nbits=%d", B, E, nbits);
    HH_REQUIRE(!bn_w || (bn_b && bn_mean && bn_var), "hash_tail_cpu: incomplete BatchNorm parameters");
    const int words = (nbits + 63) / 64;
    std::vector<float> logit((size_t)nbits);
    for (int b = 0; b < B; ++b) {
        linear(fused + (size_t)b * E, E, 1, hash_w, hash_b, nbits, E, logit.data(), nbits);
        if (packed_out) memset(packed_out + (size_t)b * words, 0, sizeof(uint64_t) * words);
        for (int j = 0; j < nbits; ++j) {
            float v = logit[(size_t)j];
            if (bn_w) v = (v - bn_mean[j]) / sqrtf(bn_var[j] + bn_eps) * bn_w[j] + bn_b[j];
            if (logits_out) logits_out[(size_t)b * nbits + j] = v;
            if (codes_out) codes_out[(size_t)b * nbits + j] = v > 0.f ? 1.f : (v < 0.f ? -1.f : (v == 0.f ? 0.f : v));
            if (packed_out && v > 0.f) packed_out[(size_t)b * words + j / 64] |= 1ull << (j % 64);
        }
    }
    return WV_OK;
}

// Host twins of the ranking-side entry points ("_cpu twins of each"; BASELINE config c0 is the CPU plumbing
// case): bit packing, per-bit counts, Hamming distances, the stable counting-sort ranking, average precision and running
// hit counts on HOST pointers.  The reference's calculator runs on CPU tensors
// (/root/reference/main/engine/accuracy_calculator.py:279-349, self.device = cpu, main/engine/evaluate.py:76-81); with
// CustomCalculator(device='cpu') -- explicit, never a silent fallback -- wvhash's does too, through these.
//
// Same results as the gfx950 kernels, bit for bit: integers by construction (packed words, counts, distances, the order
// ascending distance then ascending database row = torch.argsort(stable=True)); the average precision because the fp32
// quotients j / rank are summed in k_map_at_k's order (topk.hip: position p belongs to thread p % 256, a thread's
// quotients accumulate in a double, the 64 lanes of a wave combine in the xor-butterfly of wave_sum_f64, the four waves
// add up in index order).  No HIP call, no thread, no global state; plain C++ for the host, no fused multiply-add.
// Product code: shares nothing with oracle/.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/wvhash.h"

namespace wv {
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
}

#define HR_FAIL(code, ...)            \
    do {                              \
        ::wv::set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)
#define HR_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) HR_FAIL(WV_EINVAL, __VA_ARGS__); \
    } while (0)

namespace {

// distance of one query to N codes: the popcount instruction when the CPU has it (run-time dispatch, same integers)
template <int WORDS>
static inline void dist_row_impl(const uint64_t *q, const uint64_t *db, int64_t N, int words, uint8_t *d)
{
    const int w = WORDS > 0 ? WORDS : words;
    for (int64_t n = 0; n < N; ++n) {
        int c = 0;
        for (int j = 0; j < w; ++j) c += __builtin_popcountll(q[j] ^ db[n * w + j]);
        d[n] = (uint8_t)c;
    }
}
__attribute__((target("popcnt"))) void dist_row_popcnt(const uint64_t *q, const uint64_t *db, int64_t N, int words, uint8_t *d)
{
    if (words == 1) dist_row_impl<1>(q, db, N, words, d);
    else if (words == 2) dist_row_impl<2>(q, db, N, words, d);
    else dist_row_impl<0>(q, db, N, words, d);
}
void dist_row_plain(const uint64_t *q, const uint64_t *db, int64_t N, int words, uint8_t *d)
{
    dist_row_impl<0>(q, db, N, words, d);
}
void dist_row(const uint64_t *q, const uint64_t *db, int64_t N, int words, uint8_t *d)
{
    static const int has = __builtin_cpu_supports("popcnt") ? 1 : 0;
    if (has) dist_row_popcnt(q, db, N, words, d);
    else dist_row_plain(q, db, N, words, d);
}

inline bool relevant(const uint64_t *ql, const uint64_t *dl, int lwords)
{
    uint64_t any = 0;
    for (int w = 0; w < lwords; ++w) any |= ql[w] & dl[w];
    return any != 0;
}

}  // namespace

extern "C" int wv_pack_bits_cpu(const float *src, int64_t ld_src, uint64_t *packed, int64_t rows, int nbits, int mode,
                                int32_t *bad_flag)
{
    HR_REQUIRE(src && packed, "pack_bits_cpu: null buffer");
    HR_REQUIRE(rows >= 0 && nbits >= 1 && ld_src >= nbits, "pack_bits_cpu: bad shape rows=%lld nbits=%d ld=%lld", (long long)rows,
               nbits, (long long)ld_src);
    HR_REQUIRE(mode == 0 || mode == 1, "pack_bits_cpu: mode %d (0 = codes, 1 = labels)", mode);
    const int words = (nbits + 63) / 64;
    bool bad = false;
    for (int64_t r = 0; r < rows; ++r) {
        const float *row = src + r * ld_src;
        for (int w = 0; w < words; ++w) {
            uint64_t word = 0;
            const int n = nbits - 64 * w < 64 ? nbits - 64 * w : 64;
            for (int j = 0; j < n; ++j) {
                const float v = row[64 * w + j];
                bad |= mode == 0 ? !(v == 1.0f || v == -1.0f) : !(v >= 0.0f);
                word |= (uint64_t)(v > 0.0f) << j;
            }
            packed[r * words + w] = word;
        }
    }
    if (bad_flag && bad) *bad_flag |= 1;
    return WV_OK;
}

extern "C" int wv_bit_counts_cpu(const uint64_t *packed, int64_t rows, int nbits, uint32_t *counts)
{
    HR_REQUIRE(packed && counts, "bit_counts_cpu: null buffer");
    HR_REQUIRE(rows >= 0 && nbits >= 1, "bit_counts_cpu: bad shape");
    const int words = (nbits + 63) / 64;
    memset(counts, 0, sizeof(uint32_t) * (size_t)nbits);
    for (int64_t r = 0; r < rows; ++r)
        for (int j = 0; j < nbits; ++j) counts[j] += (uint32_t)((packed[r * words + (j >> 6)] >> (j & 63)) & 1ull);
    return WV_OK;
}

extern "C" int wv_hamming_dist_cpu(const uint64_t *q, const uint64_t *db, uint8_t *dist, int64_t ld_dist, int Q, int64_t N,
                                   int words)
{
    HR_REQUIRE(q && db && dist, "hamming_dist_cpu: null buffer");
    HR_REQUIRE(Q >= 0 && N >= 0 && ld_dist >= N, "hamming_dist_cpu: bad shape Q=%d N=%lld ld=%lld", Q, (long long)N,
               (long long)ld_dist);
    HR_REQUIRE(words >= 1 && words <= 3, "hamming_dist_cpu: %d words (uint8 distances need <= 255 bits)", words);
    for (int qi = 0; qi < Q; ++qi) dist_row(q + (int64_t)qi * words, db, N, words, dist + (int64_t)qi * ld_dist);
    return WV_OK;
}

// The k nearest rows of every query, ascending (distance, row): a counting sort over the nbits + 1 possible distances --
// O(N) per query like the kernel (the reference: an O(N log N) comparison sort of a key with <= nbits + 1 values).
extern "C" int wv_hamming_topk_cpu(const uint64_t *q, const uint64_t *db, int32_t *idx, uint8_t *dist, int Q, int64_t N,
                                   int nbits, int k, int64_t idx_offset)
{
    HR_REQUIRE(q && db && idx, "hamming_topk_cpu: null buffer");
    HR_REQUIRE(nbits >= 1 && nbits <= 128, "hamming_topk_cpu: nbits=%d (supported: 1..128)", nbits);
    HR_REQUIRE(Q >= 0 && N >= 1 && k >= 1 && k <= N, "hamming_topk_cpu: bad shape Q=%d N=%lld k=%d", Q, (long long)N, k);
    HR_REQUIRE(N + idx_offset <= 0x7fffffffLL, "hamming_topk_cpu: indices exceed int32");
    const int words = (nbits + 63) / 64, nbins = nbits + 1;
    std::vector<uint8_t> d((size_t)N);
    std::vector<int64_t> start((size_t)nbins + 1);
    for (int qi = 0; qi < Q; ++qi) {
        dist_row(q + (int64_t)qi * words, db, N, words, d.data());
        for (int b = 0; b <= nbins; ++b) start[b] = 0;
        for (int64_t n = 0; n < N; ++n) start[d[n] + 1]++;
        for (int b = 0; b < nbins; ++b) start[b + 1] += start[b];          // start[b] = rows with distance < b
        int32_t *out = idx + (int64_t)qi * k;
        uint8_t *dout = dist ? dist + (int64_t)qi * k : nullptr;
        for (int64_t n = 0; n < N; ++n) {                                   // ascending row inside a bin: stable
            const int64_t pos = start[d[n]]++;
            if (pos < k) {
                out[pos] = (int32_t)(n + idx_offset);
                if (dout) dout[pos] = d[n];
            }
        }
    }
    return WV_OK;
}

// Average precision over the first k entries of each ranked list (row pitch ld >= k), the arithmetic and the summation
// order of k_map_at_k (see the file header); entries < 0 are skipped like there.
extern "C" int wv_map_at_k_cpu(const int32_t *idx, int64_t ld, int Q, int k, const uint64_t *qlab, const uint64_t *dblab,
                               int lwords, float *ap, int32_t *nrel)
{
    HR_REQUIRE(idx && qlab && dblab && ap, "map_at_k_cpu: null buffer");
    HR_REQUIRE(Q >= 0 && k >= 1 && ld >= k && lwords >= 1, "map_at_k_cpu: bad shape Q=%d k=%d ld=%lld lwords=%d", Q, k,
               (long long)ld, lwords);
    for (int qi = 0; qi < Q; ++qi) {
        const int32_t *list = idx + (int64_t)qi * ld;
        const uint64_t *ql = qlab + (int64_t)qi * lwords;
        double acc[256];
        for (int t = 0; t < 256; ++t) acc[t] = 0.0;
        uint32_t hits = 0;
        for (int p = 0; p < k; ++p) {
            const int32_t id = list[p];
            if (id >= 0 && relevant(ql, dblab + (int64_t)id * lwords, lwords)) {
                ++hits;                                                     // the j-th hit, at rank p + 1
                acc[p & 255] += (double)((float)hits / (float)(p + 1));    // fp32 quotient like the reference
            }
        }
        double wave[4];
        for (int w = 0; w < 4; ++w) {                                       // wave_sum_f64: v += shfl_xor(v, d), d = 32 .. 1
            double s[64];
            for (int l = 0; l < 64; ++l) s[l] = acc[64 * w + l];
            for (int dd = 32; dd > 0; dd >>= 1) {
                double n2[64];
                for (int l = 0; l < 64; ++l) n2[l] = s[l] + s[l ^ dd];
                for (int l = 0; l < 64; ++l) s[l] = n2[l];
            }
            wave[w] = s[0];
        }
        const double total = wave[0] + wave[1] + wave[2] + wave[3];
        ap[qi] = hits ? (float)(total / (double)hits) : 0.0f;
        if (nrel) nrel[qi] = (int32_t)hits;
    }
    return WV_OK;
}

extern "C" int wv_hit_prefix_cpu(const int32_t *idx, int Q, int k, const uint64_t *qlab, const uint64_t *dblab, int lwords,
                                 uint32_t *hits)
{
    HR_REQUIRE(idx && qlab && dblab && hits, "hit_prefix_cpu: null buffer");
    HR_REQUIRE(Q >= 0 && k >= 1 && lwords >= 1, "hit_prefix_cpu: bad shape Q=%d k=%d lwords=%d", Q, k, lwords);
    for (int qi = 0; qi < Q; ++qi) {
        const int32_t *list = idx + (int64_t)qi * k;
        const uint64_t *ql = qlab + (int64_t)qi * lwords;
        uint32_t running = 0;
        for (int p = 0; p < k; ++p) {
            const int32_t id = list[p];
            if (id >= 0 && relevant(ql, dblab + (int64_t)id * lwords, lwords)) ++running;
            hits[(int64_t)qi * k + p] = running;
        }
    }
    return WV_OK;
}

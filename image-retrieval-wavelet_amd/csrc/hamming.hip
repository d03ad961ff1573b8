// Bit packing, per-bit counts and the Hamming distance matrix for gfx950.
//
// Reference semantics (accuracy_calculator.py:183-186): dist = 0.5 * (B - q @ r.T) on fp32
// +-1 codes.  Here codes are packed 64 per word (bit = value > 0), so the distance is
// popcount(q ^ r): 8 bytes per 64-bit code instead of 256, and an exact integer.
//
// k_hamming_dist is the kernel the HBM roofline target is quoted on: per launch it must write
// Q*N bytes and read (Q+N)*B/8.  Database codes are staged per workgroup through LDS so that
// every lane ends up owning CPT *consecutive* codes (-> one 16-byte store per query row) while the
// global loads stay fully coalesced; each staged tile is reused for QCH queries held in SGPRs.
#include "common.hpp"

namespace wv {

// ---------------------------------------------------------------------------------- packing
// one wave packs one 64-bit word per step: lane j reads element 64*w + j, ballot builds the word
__global__ __launch_bounds__(256) void k_pack_bits(const float *__restrict__ src, int64_t ld,
                                                   uint64_t *__restrict__ packed, int64_t rows,
                                                   int nbits, int words, int mode,
                                                   int32_t *__restrict__ bad_flag)
{
    const int lane = lane_id();
    const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_id();
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t total = rows * words;
    bool bad = false;
    for (int64_t item = wave_global; item < total; item += nwaves) {
        const int64_t row = item / words;
        const int w = (int)(item - row * words);
        const int col = w * 64 + lane;
        float v = 0.f;
        const bool in_range = col < nbits;
        if (in_range) v = src[row * ld + col];
        if (in_range) {
            if (mode == 0) bad |= !(v == 1.0f || v == -1.0f);
            else bad |= !(v >= 0.0f);
        }
        const uint64_t word = __ballot(in_range && v > 0.0f);
        if (lane == 0) packed[item] = word;
    }
    if (bad_flag && __any(bad)) {
        if (lane == 0) atomicOr(bad_flag, 1);
    }
}

// counts[j] = number of rows with bit j set
__global__ __launch_bounds__(256) void k_bit_counts(const uint64_t *__restrict__ packed,
                                                    int64_t rows, int words, int nbits,
                                                    uint32_t *__restrict__ counts)
{
    const int lane = lane_id();
    const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_id();
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t chunks = (rows + 63) / 64;
    for (int w = 0; w < words; ++w) {
        uint32_t mine = 0;  // lane j accumulates bit j of word w
        for (int64_t ch = wave_global; ch < chunks; ch += nwaves) {
            const int64_t row = ch * 64 + lane;
            const uint64_t word = row < rows ? packed[row * words + w] : 0ull;
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const uint32_t c = __popcll(__ballot((word >> j) & 1ull));
                if (lane == j) mine += c;
            }
        }
        const int bit = w * 64 + lane;
        if (bit < nbits && mine) atomicAdd(&counts[bit], mine);
    }
}

// --------------------------------------------------------------------------- distance matrix
template <int WORDS>
struct DistCfg {
    static constexpr int CPT = WORDS == 1 ? 16 : (WORDS == 2 ? 8 : 4);  // codes (= out bytes) per thread
    static constexpr int TILE = 256 * CPT;                                // codes per workgroup
    static constexpr int CODE_BYTES = WORDS * 8;
    static constexpr int THREAD_BYTES = CPT * CODE_BYTES;                 // 128 for WORDS 1,2,4; 96 for 3
    static constexpr int THREAD_PITCH = THREAD_BYTES + 16;                // +16 B: conflict-free b128 reads
};

template <int WORDS, bool ALIGNED>
__global__ __launch_bounds__(256) void k_hamming_dist(const uint64_t *__restrict__ q,
                                                      const uint64_t *__restrict__ db,
                                                      uint8_t *__restrict__ dist, int64_t ld,
                                                      int Q, int64_t N, int qch)
{
    using Cfg = DistCfg<WORDS>;
    constexpr int CPT = Cfg::CPT;
    __shared__ uint4 stage4[256 * Cfg::THREAD_PITCH / 16];
    uint8_t *stage = reinterpret_cast<uint8_t *>(stage4);
    const int tid = threadIdx.x;
    const int64_t n_tile = (int64_t)blockIdx.x * Cfg::TILE;
    const int q0 = blockIdx.y * qch;
    const int q1 = min(q0 + qch, Q);

    // coalesced 16-byte loads of the tile; LDS image is [owner thread][its CPT codes] (+pad)
    {
        constexpr int CHUNKS = Cfg::TILE * Cfg::CODE_BYTES / 16;           // 16-B chunks in the tile
        constexpr int PER_THREAD = Cfg::THREAD_BYTES / 16;                 // chunks owned per thread
        const int64_t tile_bytes_valid = (min((int64_t)Cfg::TILE, N - n_tile)) * Cfg::CODE_BYTES;
        const uint8_t *gsrc = reinterpret_cast<const uint8_t *>(db) + n_tile * Cfg::CODE_BYTES;
#pragma unroll
        for (int i = 0; i < CHUNKS / 256; ++i) {
            const int ch = i * 256 + tid;
            const int64_t off = (int64_t)ch * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (off + 16 <= tile_bytes_valid) {
                v = *reinterpret_cast<const uint4 *>(gsrc + off);
            } else if (off < tile_bytes_valid) {  // WORDS odd: last valid code straddles a chunk
                const uint64_t lo = *reinterpret_cast<const uint64_t *>(gsrc + off);
                v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32);
            }
            const int owner = ch / PER_THREAD, slot = ch - owner * PER_THREAD;
            *reinterpret_cast<uint4 *>(stage + owner * Cfg::THREAD_PITCH + slot * 16) = v;
        }
    }
    __syncthreads();
    uint64_t code[CPT][WORDS];
    {
        const uint8_t *mine = stage + tid * Cfg::THREAD_PITCH;
        constexpr int PER_THREAD = Cfg::THREAD_BYTES / 16;
        uint64_t flat[PER_THREAD * 2];
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            uint4 v = *reinterpret_cast<const uint4 *>(mine + i * 16);
            flat[2 * i] = (uint64_t)v.x | ((uint64_t)v.y << 32);
            flat[2 * i + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c)
#pragma unroll
            for (int w = 0; w < WORDS; ++w) code[c][w] = flat[c * WORDS + w];
    }
    const int64_t n0 = n_tile + (int64_t)tid * CPT;
    if (n0 >= N) return;
    const bool full = n0 + CPT <= N;

    for (int qi = q0; qi < q1; ++qi) {
        uint64_t qw[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; ++w) qw[w] = q[(int64_t)qi * WORDS + w];  // uniform -> scalar loads
        uint32_t outw[CPT / 4];
#pragma unroll
        for (int g4 = 0; g4 < CPT / 4; ++g4) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < WORDS; ++w) d += __popcll(code[g4 * 4 + j][w] ^ qw[w]);
                acc |= d << (8 * j);
            }
            outw[g4] = acc;
        }
        uint8_t *o = dist + (int64_t)qi * ld + n0;
        if (ALIGNED && full) {
            if constexpr (CPT == 16)
                *reinterpret_cast<uint4 *>(o) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
            else if constexpr (CPT == 8)
                *reinterpret_cast<uint2 *>(o) = make_uint2(outw[0], outw[1]);
            else
                *reinterpret_cast<uint32_t *>(o) = outw[0];
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (n0 + c < N) o[c] = (uint8_t)(outw[c / 4] >> (8 * (c & 3)));
        }
    }
}

// ---- prepared database: the tile image k_hamming_dist stages through LDS, stored once in HBM.
// dbP[tile][j][t] (16 B) = bytes [t * THREAD_BYTES + 16 j, +16) of tile `tile` of the natural array, so that
// thread t's CPT consecutive codes arrive with PER_THREAD fully coalesced 16-byte loads and no LDS pass.
template <int WORDS>
__global__ __launch_bounds__(256) void k_db_permute(const uint64_t *__restrict__ db, uint4 *__restrict__ dbP,
                                                    int64_t N, int64_t tiles)
{
    using Cfg = DistCfg<WORDS>;
    constexpr int PT = Cfg::THREAD_BYTES / 16;
    const int64_t total = tiles * PT * 256;
    const int64_t valid_bytes = N * Cfg::CODE_BYTES;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(db);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i & 255);
        const int64_t tj = i >> 8;
        const int j = (int)(tj % PT);
        const int64_t tile = tj / PT;
        const int64_t off = tile * Cfg::TILE * Cfg::CODE_BYTES + (int64_t)t * Cfg::THREAD_BYTES + 16 * j;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (off + 16 <= valid_bytes) {
            v = *reinterpret_cast<const uint4 *>(src + off);
        } else if (off < valid_bytes) {
            const uint64_t lo = *reinterpret_cast<const uint64_t *>(src + off);
            v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32);
        }
        dbP[i] = v;
    }
}

template <int WORDS, bool ALIGNED, int QCH>
__global__ __launch_bounds__(256) void k_hamming_dist_prep(const uint64_t *__restrict__ q,
                                                           const uint4 *__restrict__ dbP,
                                                           uint8_t *__restrict__ dist, int64_t ld, int Q,
                                                           int64_t N)
{
    using Cfg = DistCfg<WORDS>;
    constexpr int CPT = Cfg::CPT, PT = Cfg::THREAD_BYTES / 16;
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * Cfg::TILE + (int64_t)tid * CPT;
    const int q0 = blockIdx.y * QCH;
    uint64_t flat[PT * 2];
    const uint4 *src = dbP + ((int64_t)blockIdx.x * PT) * 256 + tid;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const uint4 v = src[j * 256];
        flat[2 * j] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        flat[2 * j + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    if (n0 >= N) return;
    const bool full = n0 + CPT <= N;
#pragma unroll
    for (int qq = 0; qq < QCH; ++qq) {
        const int qi = q0 + qq;
        if (qi >= Q) break;
        uint64_t qw[WORDS];
#pragma unroll
        for (int w = 0; w < WORDS; ++w) qw[w] = q[(int64_t)qi * WORDS + w];  // uniform -> scalar loads
        uint32_t outw[CPT / 4];
#pragma unroll
        for (int g4 = 0; g4 < CPT / 4; ++g4) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < WORDS; ++w) d += __popcll(flat[(g4 * 4 + j) * WORDS + w] ^ qw[w]);
                acc |= d << (8 * j);
            }
            outw[g4] = acc;
        }
        uint8_t *o = dist + (int64_t)qi * ld + n0;
        if (ALIGNED && full) {
            if constexpr (CPT == 16)
                *reinterpret_cast<uint4 *>(o) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
            else if constexpr (CPT == 8)
                *reinterpret_cast<uint2 *>(o) = make_uint2(outw[0], outw[1]);
            else
                *reinterpret_cast<uint32_t *>(o) = outw[0];
        } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                if (n0 + c < N) o[c] = (uint8_t)(outw[c / 4] >> (8 * (c & 3)));
        }
    }
}

template <int WORDS, int QCH>
static int launch_dist_prep_q(const uint64_t *q, const void *dbP, uint8_t *dist, int64_t ld, int Q, int64_t N,
                              hipStream_t st)
{
    using Cfg = DistCfg<WORDS>;
    const int64_t tiles = ceil_div(N, Cfg::TILE);
    const int64_t qblocks = ceil_div(Q, QCH);
    if (tiles > 0x7fffffff || qblocks > 65535) WV_FAIL(WV_ENOTSUP, "hamming_dist: grid too large");
    const bool aligned = (ld % 16 == 0) && ((reinterpret_cast<uintptr_t>(dist) & 15) == 0);
    dim3 grid((unsigned)tiles, (unsigned)qblocks);
    if (aligned)
        hipLaunchKernelGGL((k_hamming_dist_prep<WORDS, true, QCH>), grid, dim3(256), 0, st, q, (const uint4 *)dbP, dist, ld, Q, N);
    else
        hipLaunchKernelGGL((k_hamming_dist_prep<WORDS, false, QCH>), grid, dim3(256), 0, st, q, (const uint4 *)dbP, dist, ld, Q, N);
    WV_CHECK_LAUNCH("k_hamming_dist_prep");
    return WV_OK;
}

template <int WORDS>
static int launch_dist_prep(const uint64_t *q, const void *dbP, uint8_t *dist, int64_t ld, int Q, int64_t N,
                            hipStream_t st)
{
    using Cfg = DistCfg<WORDS>;
    // queries per workgroup: enough workgroups for ~2 generations per CU slot, so that the loads of one
    // generation overlap the output stream of the previous one
    int qch = 4;   // measured at 16 384 queries x 25 000 codes: 4.00 / 4.61 / 4.45 / 3.82 TB/s for 2 / 4 / 8 / 16 queries per workgroup
    const int64_t tiles = ceil_div(N, Cfg::TILE);
    while (qch > 2 && tiles * ceil_div(Q, qch) < 4096) qch >>= 1;
    while (qch < 16 && ceil_div(Q, qch) > 65535) qch <<= 1;
    if (const char *e = ::wv::tune("WV_DIST_QCH")) qch = atoi(e);
    switch (qch) {
    case 2: return launch_dist_prep_q<WORDS, 2>(q, dbP, dist, ld, Q, N, st);
    case 4: return launch_dist_prep_q<WORDS, 4>(q, dbP, dist, ld, Q, N, st);
    case 16: return launch_dist_prep_q<WORDS, 16>(q, dbP, dist, ld, Q, N, st);
    default: return launch_dist_prep_q<WORDS, 8>(q, dbP, dist, ld, Q, N, st);
    }
}

template <int WORDS>
static int launch_permute(const uint64_t *db, void *dbP, int64_t N, hipStream_t st)
{
    using Cfg = DistCfg<WORDS>;
    const int64_t tiles = ceil_div(N, Cfg::TILE);
    const int64_t total = tiles * (Cfg::THREAD_BYTES / 16) * 256;
    hipLaunchKernelGGL((k_db_permute<WORDS>), dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096)), dim3(256), 0,
                       st, db, (uint4 *)dbP, N, tiles);
    WV_CHECK_LAUNCH("k_db_permute");
    return WV_OK;
}

size_t dist_prepared_bytes(int64_t N, int words)
{
    switch (words) {
    case 1: return (size_t)ceil_div(N, DistCfg<1>::TILE) * DistCfg<1>::TILE * DistCfg<1>::CODE_BYTES;
    case 2: return (size_t)ceil_div(N, DistCfg<2>::TILE) * DistCfg<2>::TILE * DistCfg<2>::CODE_BYTES;
    case 3: return (size_t)ceil_div(N, DistCfg<3>::TILE) * DistCfg<3>::TILE * DistCfg<3>::CODE_BYTES;
    default: return (size_t)ceil_div(N, DistCfg<4>::TILE) * DistCfg<4>::TILE * DistCfg<4>::CODE_BYTES;
    }
}

int dist_prepare(const uint64_t *db, void *dbP, int64_t N, int words, hipStream_t st)
{
    switch (words) {
    case 1: return launch_permute<1>(db, dbP, N, st);
    case 2: return launch_permute<2>(db, dbP, N, st);
    case 3: return launch_permute<3>(db, dbP, N, st);
    default: return launch_permute<4>(db, dbP, N, st);
    }
}

template <int WORDS>
static int launch_dist(const uint64_t *q, const uint64_t *db, uint8_t *dist, int64_t ld, int Q,
                       int64_t N, hipStream_t st)
{
    using Cfg = DistCfg<WORDS>;
    const int64_t tiles = ceil_div(N, Cfg::TILE);
    // enough workgroups to fill 256 CUs several times over, but keep >= 8 queries per staged tile
    int qch = 32;
    while (qch > 8 && tiles * ceil_div(Q, qch) < 2048) qch >>= 1;
    if (const char *e = ::wv::tune("WV_DIST_QCH")) qch = std::max(1, atoi(e));
    const int64_t qblocks = ceil_div(Q, qch);
    if (tiles > 0x7fffffff || qblocks > 65535) WV_FAIL(WV_ENOTSUP, "hamming_dist: grid too large");
    const bool aligned = (ld % 16 == 0) && ((reinterpret_cast<uintptr_t>(dist) & 15) == 0);
    dim3 grid((unsigned)tiles, (unsigned)qblocks);
    if (aligned)
        hipLaunchKernelGGL((k_hamming_dist<WORDS, true>), grid, dim3(256), 0, st, q, db, dist, ld, Q, N, qch);
    else
        hipLaunchKernelGGL((k_hamming_dist<WORDS, false>), grid, dim3(256), 0, st, q, db, dist, ld, Q, N, qch);
    WV_CHECK_LAUNCH("k_hamming_dist");
    return WV_OK;
}

}  // namespace wv

using namespace wv;

extern "C" int wv_pack_bits(const float *src, int64_t ld_src, uint64_t *packed, int64_t rows,
                            int nbits, int mode, int32_t *bad_flag, void *stream)
{
    WV_REQUIRE(src && packed, "pack_bits: null buffer");
    WV_REQUIRE(rows >= 0 && nbits >= 1 && ld_src >= nbits, "pack_bits: bad shape rows=%lld nbits=%d ld=%lld",
               (long long)rows, nbits, (long long)ld_src);
    WV_REQUIRE(mode == 0 || mode == 1, "pack_bits: mode %d", mode);
    if (rows == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    const int64_t total = rows * words;
    const int grid = (int)std::min<int64_t>(ceil_div(total, 4), 256 * 8);
    hipLaunchKernelGGL(k_pack_bits, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, packed,
                       rows, nbits, words, mode, bad_flag);
    WV_CHECK_LAUNCH("k_pack_bits");
    return WV_OK;
}

extern "C" int wv_bit_counts(const uint64_t *packed, int64_t rows, int nbits, uint32_t *counts,
                             void *stream)
{
    WV_REQUIRE(packed && counts, "bit_counts: null buffer");
    WV_REQUIRE(rows >= 0 && nbits >= 1, "bit_counts: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(uint32_t) * nbits, st);
    if (e != hipSuccess) WV_FAIL(WV_EHIP, "bit_counts: memset: %s", hipGetErrorString(e));
    if (rows == 0) return WV_OK;
    const int words = (nbits + 63) / 64;
    const int grid = (int)std::min<int64_t>(ceil_div(ceil_div(rows, 64), 4), 256);
    hipLaunchKernelGGL(k_bit_counts, dim3(grid), dim3(256), 0, st, packed, rows, words, nbits, counts);
    WV_CHECK_LAUNCH("k_bit_counts");
    return WV_OK;
}

extern "C" int wv_hamming_dist(const uint64_t *q, const uint64_t *db, uint8_t *dist, int64_t ld_dist,
                               int Q, int64_t N, int words, void *stream)
{
    WV_REQUIRE(q && db && dist, "hamming_dist: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 0 && ld_dist >= N, "hamming_dist: bad shape Q=%d N=%lld ld=%lld", Q,
               (long long)N, (long long)ld_dist);
    WV_REQUIRE(words >= 1 && words <= 4, "hamming_dist: words=%d (nbits must be <= 255)", words);
    if (Q == 0 || N == 0) return WV_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (words) {
    case 1: return launch_dist<1>(q, db, dist, ld_dist, Q, N, st);
    case 2: return launch_dist<2>(q, db, dist, ld_dist, Q, N, st);
    case 3: return launch_dist<3>(q, db, dist, ld_dist, Q, N, st);
    default: return launch_dist<4>(q, db, dist, ld_dist, Q, N, st);
    }
}

extern "C" int wv_hamming_dist_prepared(const uint64_t *q, const void *prepared, uint8_t *dist, int64_t ld_dist,
                                        int Q, int64_t N, int words, void *stream)
{
    WV_REQUIRE(q && prepared && dist, "hamming_dist_prepared: null buffer");
    WV_REQUIRE(Q >= 0 && N >= 0 && ld_dist >= N, "hamming_dist_prepared: bad shape Q=%d N=%lld ld=%lld", Q,
               (long long)N, (long long)ld_dist);
    WV_REQUIRE(words >= 1 && words <= 4, "hamming_dist_prepared: words=%d (nbits must be <= 255)", words);
    if (Q == 0 || N == 0) return WV_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (words) {
    case 1: return launch_dist_prep<1>(q, prepared, dist, ld_dist, Q, N, st);
    case 2: return launch_dist_prep<2>(q, prepared, dist, ld_dist, Q, N, st);
    case 3: return launch_dist_prep<3>(q, prepared, dist, ld_dist, Q, N, st);
    default: return launch_dist_prep<4>(q, prepared, dist, ld_dist, Q, N, st);
    }
}

// placeholder until the real kernels land later this round
#include "common.hpp"
extern "C" size_t wv_band_attn_pool_workspace_bytes(const wv_head_params *, int) { return 0; }
extern "C" int wv_band_attn_pool(const wv_head_params *, const float *, int, float *, void *, size_t, void *)
{
    WV_FAIL(WV_ENOTSUP, "band_attn_pool: not built yet");
}
extern "C" int wv_hash_tail(const float *, int, int, const float *, const float *, const float *, const float *,
                            const float *, const float *, float, int, float *, float *, uint64_t *, void *)
{
    WV_FAIL(WV_ENOTSUP, "hash_tail: not built yet");
}

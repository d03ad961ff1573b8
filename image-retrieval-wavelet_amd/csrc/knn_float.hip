// placeholder until the real kernels land later this round
#include "common.hpp"
extern "C" size_t wv_knn_float_workspace_bytes(int, int64_t, int, int) { return 0; }
extern "C" int wv_knn_float(const float *, const float *, int, int64_t, int, int, int, int32_t *, float *,
                            void *, size_t, void *)
{
    WV_FAIL(WV_ENOTSUP, "knn_float: not built yet");
}

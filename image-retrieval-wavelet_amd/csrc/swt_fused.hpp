// Entry points of the register-fused two-pass SWT kernel (swt_fused.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace wv {
// true when the (taps, levels, width) combination has a fused instantiation
bool swt_fused_covers(int L, int n, int W);
// returns WV_OK / negative error, or 1 when the shape is not covered
int swt_fused_launch(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B, int C, int H,
                     int W, int n, const float *lo, const float *hi, int L, hipStream_t st);
// sliding-window persistent kernel (swt_slide.hip): full-width rows, returns 1 when not covered
bool swt_slide_covers(int L, int n, int W, int H);
// out_layout WV_BANDS_OUTER: out is [4][..][C][H][W] with `band_stride` elements between the bands of one plane
int swt_slide_launch(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B, int C, int H,
                     int W, int n, const float *lo, const float *hi, int L, hipStream_t st, int out_layout = 0,
                     int64_t band_stride = 0);
}  // namespace wv

// Bin geometry of position-sensitive RoI pooling, shared by the plane-stationary kernels (psroi.hip) and the
// position-major kernels (heads.hip).  Follows psroi_pooling_kernel.cu:29-59 operation by operation (double where the
// reference's literals promote to double, no FMA contraction), so bin boundaries match the oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>

struct Bin { int hstart, hend, wstart, wend; bool empty; };

// psroi_pooling_kernel.cu:29-59
__device__ __forceinline__ Bin psroi_bin(const float* __restrict__ roi, float spatial_scale, int ph, int pw,
                                         int pooled_height, int pooled_width, int height, int width) {
  const float roi_start_w = (float)round((double)roi[1]) * spatial_scale;
  const float roi_start_h = (float)round((double)roi[2]) * spatial_scale;
  const float roi_end_w = (float)(round((double)roi[3]) + 1.) * spatial_scale;
  const float roi_end_h = (float)(round((double)roi[4]) + 1.) * spatial_scale;
  const double dw = (double)(roi_end_w - roi_start_w), dh = (double)(roi_end_h - roi_start_h);
  const float roi_width = (float)(dw > 0.1 ? dw : 0.1);  // max(float, 0.1): double compare, then narrowed
  const float roi_height = (float)(dh > 0.1 ? dh : 0.1);
  const float bin_size_h = roi_height / (float)pooled_height;
  const float bin_size_w = roi_width / (float)pooled_width;
  Bin b;
  b.hstart = (int)floorf((float)ph * bin_size_h + roi_start_h);
  b.wstart = (int)floorf((float)pw * bin_size_w + roi_start_w);
  b.hend = (int)ceilf((float)(ph + 1) * bin_size_h + roi_start_h);
  b.wend = (int)ceilf((float)(pw + 1) * bin_size_w + roi_start_w);
  b.hstart = min(max(b.hstart, 0), height);
  b.hend = min(max(b.hend, 0), height);
  b.wstart = min(max(b.wstart, 0), width);
  b.wend = min(max(b.wend, 0), width);
  b.empty = (b.hend <= b.hstart) || (b.wend <= b.wstart);
  return b;
}

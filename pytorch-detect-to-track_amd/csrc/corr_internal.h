// Internal interface between the channels-last correlation entry point (correlation_nhwc.hip) and the window-split kernel
// (correlation_wsplit.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

// kernel_size 1, stride1 == stride2 = s, displacement and padding multiples of s, window radius 1 .. 16, ic % 16 == 0
int dtt_corr_wsplit_supported(int ic, int kernel_size, int max_displacement, int pad_size, int stride1, int stride2);
// max_workgroups: 0 = one round over every CU; n = plan for n CUs (the caller runs other kernels beside this one)
int dtt_corr_wsplit_forward(float* output, int ob, int oc, int oh, int ow, long out_batch_stride, long out_ch_stride,
                            long out_px_stride, const float* input1, int ic, int ih, int iw, const float* input2,
                            int pad_size, int max_displacement, int stride, int max_workgroups, hipStream_t stream);

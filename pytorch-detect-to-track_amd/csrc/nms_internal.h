// Internal (non-ABI) entry points of nms.hip shared with proposal.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

size_t dtt_nms_mask_bytes(int boxes_num);

// boxes: [batch] x (n, boxes_dim) rows sorted by descending score; n_per_image may be NULL (= n_max).
// mask: [batch] x (n_max, ceil(n_max/64)) uint64 scratch.  keep_out [batch] x keep_batch_stride,
// num_out [batch].  rois_out (optional): [batch, rois_rows, 5] written by the sweep epilogue.
int dtt_nms_batched_launch(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image,
                           int n_max, int batch, float thresh, int max_keep, unsigned long long* mask,
                           long mask_batch_stride, int* keep_out, long keep_batch_stride, int* num_out,
                           float* rois_out, int rois_rows, hipStream_t stream);

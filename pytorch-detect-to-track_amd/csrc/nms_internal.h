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

// The same in two stream-ordered halves, for a caller that produces the boxes in two halves as well (proposal.hip decodes
// the first split * 1024 boxes, runs phase 1, decodes the rest, runs phase 2): split = dtt_nms_split(...) super-chunks of
// 1024 boxes in phase 1 (0 = phase 1 is the whole NMS and phase 2 a no-op).  The done flag of image i after phase 1 is the
// 64-bit word mask[i * mask_batch_stride + n_max * ceil(n_max / 64)] (non-zero = finished).
int dtt_nms_split(int n_max, int max_keep, int have_keep_out);
int dtt_nms_phase1(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image, int n_max, int batch,
                   float thresh, int max_keep, unsigned long long* mask, long mask_batch_stride, int* keep_out,
                   long keep_batch_stride, int* num_out, float* rois_out, int rois_rows, int split, hipStream_t stream);
int dtt_nms_phase2(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image, int n_max, int batch,
                   float thresh, int max_keep, unsigned long long* mask, long mask_batch_stride, int* keep_out,
                   long keep_batch_stride, int* num_out, float* rois_out, int rois_rows, int split, hipStream_t stream);

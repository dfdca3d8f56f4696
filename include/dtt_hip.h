/*
 * dtt_hip.h -- C ABI of libdtt_hip.so, the MI355X (gfx950) Detect-to-Track hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one of the reference's
 * `extern "C"` CUDA launchers (the layer its cffi shims bind).  The
 * reference file:line each symbol replaces is cited next to it (paths relative to the
 * reference's lib/model/).  Argument order follows the reference launcher; where the
 * reference passed tensor strides that it never honoured (it asserts contiguity,
 * correlation/functions/correlation.py:21-22) they are dropped, and where the reference
 * allocated scratch inside the call (correlation_cuda.c:36-42, nms_cuda_kernel.cu:95-105)
 * the caller now passes a workspace instead: the library never allocates, frees or
 * synchronises.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless the name ends in _host
 *   - all tensors are contiguous fp32 NCHW / row-major, indices int32
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued on it and the call returns without waiting
 *   - return value: 1 = launched, 0 = failure (same convention as the reference launchers,
 *     e.g. correlation_cuda_kernel.cu:362-368); on 0, dtt_last_error() describes why.
 *     Unlike psroi_pooling_kernel.cu:99-103 / roi_align_kernel.cu:85-88 nothing calls exit().
 */
#ifndef DTT_HIP_H
#define DTT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- misc */
int dtt_abi_version(void);
const char* dtt_last_error(void);
/* Measurement hook (no reference counterpart): bracket every launch (or whole op) named `tag` with
 * hipEventRecord(begin_events[i]) / hipEventRecord(end_events[i]) on the launch stream, for the first n launches.
 * Tags -- ops: "corr_fwd_op" (one forward correlation, whatever kernels it takes), "corr_bwd_op" (both gradients of one
 * correlation: band kernel + the streamed launch; phase 2 alone when the op is issued in phases, whose phase 1 is "corr_bwd_band"),
 * "proposal_op" (the proposal layer of all images of a call, "nms_op" inside it: both NMS phases), "anchor_target_op",
 * "rpn_loss" (forward of both RPN losses), "psroi_pm_bwd" (the one-launch PSRoI backward of a map's heads);
 * kernels: "corr_nhwc" (corr_wsplit_kernel), "corr_fwd_mfma" /
 * "corr_fwd_reduce" (the NCHW pair), "head_gemm", "rpn_head_gemm", "psroi_pm" (position-major poolings), "psroi_fwd_plane",
 * "nms_mask", "nms_sweep", "proposal_select_sort".  tag = NULL detaches.  Events are hipEvent_t handles owned
 * by the caller.  dtt_profile_count() = launches recorded so far. */
int dtt_profile_attach(const char* tag, void** begin_events, void** end_events, int n);
int dtt_profile_count(void);

/* ---------------------------------------------------------------- correlation
 * Replaces Correlation_forward_cuda_kernel  (correlation/src/correlation_cuda_kernel.cu:296-369)
 *      and Correlation_backward_cuda_kernel (correlation/src/correlation_cuda_kernel.cu:371-473),
 * plus the output-shape rule of Correlation_forward_cuda (correlation/src/correlation_cuda.c:25-34).
 * No NHWC repack (`channels_first`, .cu:10-32) and no zero-filled padded copies: zero padding is
 * applied on the fly.  corr_type_multiply is accepted and ignored, as in the reference.
 */
/* writes nOutputChannels, outputHeight, outputWidth; returns 0 on invalid parameters */
int dtt_correlation_output_shape(int ic, int ih, int iw, int pad_size, int kernel_size,
                                 int max_displacement, int stride1, int stride2,
                                 int* oc, int* oh, int* ow);
size_t dtt_correlation_forward_workspace_bytes(int batch, int ic, int ih, int iw, int pad_size,
                                               int kernel_size, int max_displacement,
                                               int stride1, int stride2);
/* output may be a channel slice of a larger (ob, out_batch_stride/(oh*ow), oh, ow) tensor:
 * out_batch_stride is the element stride between batch items (oc*oh*ow when dense), so the three
 * correlations can write straight into the tracking concat buffer (rfcn.py:166-174). */
int dtt_correlation_forward(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                            const float* input1, int ic, int ih, int iw,
                            const float* input2,
                            void* workspace, size_t workspace_bytes,
                            int pad_size, int kernel_size, int max_displacement,
                            int stride1, int stride2, int corr_type_multiply,
                            void* stream);
/* Same, with explicit output strides: element (n, d, y, x) is written to output[n * out_batch_stride + d *
 * out_channel_stride + (y * ow + x) * out_pixel_stride].  (oh*ow, 1) is the call above; (1, ld) writes a column block of a
 * position-major (pixels, ld) matrix -- the layout dtt_head_gemm reads, so the three correlations of rfcn.py:166-174
 * land directly in the tracking head's input rows. */
int dtt_correlation_forward_strided(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                    long out_channel_stride, long out_pixel_stride,
                                    const float* input1, int ic, int ih, int iw, const float* input2,
                                    void* workspace, size_t workspace_bytes,
                                    int pad_size, int kernel_size, int max_displacement,
                                    int stride1, int stride2, int corr_type_multiply, void* stream);
/* The same forward on CHANNELS-LAST inputs: input1 / input2 are (ob, ih, iw, ic) row-major, i.e. what the reference's own
 * kernel reads after its `channels_first` repack (correlation_cuda_kernel.cu:10-32, 296-340) -- callers whose trunk is
 * channels-last skip both that repack and the NHWC -> NCHW hand-over.  Output addressing as above.  ONE launch per op and
 * no partial sums in memory (round 3, csrc/correlation_wsplit.hip): the (2R+1)^2 displacement window of a 4 x 4 pixel block
 * is dealt to `parts` wave pairs in runs of 4 x 4 window blocks and every pair runs ALL channels for its run (its two waves
 * take alternate 16-channel chunks and meet in LDS, even + odd, at the end) -- no channel slices across workgroups, no slabs,
 * no tickets, no workspace, no inter-workgroup communication of any kind: a result does not depend on how the launch was
 * partitioned, it is bit-identical from run to run by construction, and concurrent calls on different streams share nothing.
 * (Round 2's channel-split kernel with its ticketed in-launch slab reduction -- whose ordering argument leant on ISA behaviour
 * rather than on the HIP memory model -- is gone, and with it dtt_correlation_nhwc_workspace_bytes and the workspace arguments.)
 * Supported: kernel_size 1, stride1 == stride2 = s, pad and displacement multiples of s, ic % 16 == 0, window radius
 * max_displacement / s in 1 .. 16 (natively: d = 16 is one 33 x 33 window, not four sub-windows); 16-byte aligned inputs.
 * max_workgroups: 0 = plan the launch for every CU of the device (one workgroup per CU where the shape allows it: 256 at the
 * 600 px shapes); n > 0 = plan for n CUs -- the caller runs other kernels beside this one and wants neither to wait for the
 * other's CUs (the plan then uses more, shorter workgroups in several rounds).  dtt_correlation_nhwc_plan reports the plan
 * (window parts, accumulators per wave, workgroups, LDS ring slots of the largest tile) without launching anything. */
int dtt_correlation_forward_nhwc(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                 long out_channel_stride, long out_pixel_stride,
                                 const float* input1, int ic, int ih, int iw, const float* input2,
                                 int pad_size, int kernel_size, int max_displacement,
                                 int stride1, int stride2, int max_workgroups, void* stream);
int dtt_correlation_nhwc_plan(int batch, int oh, int ow, int window_radius, int max_workgroups, int* parts,
                              int* accumulators, int* workgroups, int* ring_slots);
/* Host-only self-check of that plan: replays every work item through the kernel's own item decode and returns 1 iff every
 * (image, 4 x 4 pixel block, window block) triple is owned by exactly one wave and every workgroup's halo fits its LDS-DMA budget. */
int dtt_correlation_nhwc_plan_check(int batch, int oh, int ow, int window_radius, int max_workgroups);
/* Any kernel_size / strides (D&T itself uses kernel_size 1, rfcn.py:58-60: that case runs on the matrix cores).
 * gradInput1/2 are fully written (no pre-zeroing needed).  The gradients are the mathematically exact adjoint of
 * dtt_correlation_forward.  The reference's own backward departs from its forward in two places, which are NOT
 * reproduced: for stride1 > 1 it indexes gradOutput out of bounds (correlation_cuda_kernel.cu:120-121, 212-213), and for
 * kernel_size > 1 it sums the outputs whose index lies within kernel_rad of (x - max_displacement) / stride1 (:128-132,
 * 222-226), i.e. treats the patch of its forward (anchored at its top-left corner, :48-49, 56-63) as centred. */
int dtt_correlation_backward(const float* gradOutput, int gob, int goc, int goh, int gow,
                             const float* input1, int ic, int ih, int iw,
                             const float* input2,
                             float* gradInput1, float* gradInput2,
                             int pad_size, int kernel_size, int max_displacement,
                             int stride1, int stride2, int corr_type_multiply,
                             void* stream);
/* The same gradients for channels-last inputs (gob, ih, iw, ic) and channels-last gradInput1/2 -- the layout of a channels-last
 * training trunk (dtt/ops.py: CorrelationNHWCFunction; forward = dtt_correlation_forward_nhwc with the NCHW output strides).
 * Matrix-core paths only: kernel_size 1, stride1 == stride2, max_displacement / stride <= 8 (every correlation D&T trains,
 * rfcn.py:58-60); anything else fails with an error string and the caller converts to NCHW for dtt_correlation_backward.
 * Replace Correlation_backward_input1 / _input2 (correlation_cuda_kernel.cu:108-290) and their launcher (:371-473).
 *
 * dtt_correlation_backward_nhwc_strided (ic % 64 == 0: conv3 / conv4 / conv5; dtt_correlation_backward_stream_supported):
 * band-stationary, halo-streamed kernels (csrc/correlation_bwd.hip) -- the gradient band of a 4 x 4 target block stays in
 * registers, the other frame's halo streams through an LDS-DMA ring 64 channels wide, one ds_read_b128 per four exact-f32 MFMAs,
 * float4 stores straight from the accumulators; fixed summation order, no atomics, no partial sums.  gradOut[n, d, p] is read at
 * gradOutput[n * g_batch_stride + d * g_ch_stride + p * g_px_stride], p = oy * ow + ox: contiguous planes (D*D*oh*ow, oh*ow, 1)
 * or columns of position-major rows (g_ch_stride = 1, g_px_stride = row length: the gradient of the tracking head's input
 * rows, no copy).  which: 1 = gradInput1 only, 2 = gradInput2 only, 3 = both.  workspace: caller-owned,
 * dtt_correlation_backward_workspace_bytes(...) bytes (the band words in register order, written by a small first launch).
 *
 * dtt_correlation_backward_nhwc (ic % 16 == 0): round 1's kernel in its channels-last instantiation; gradOutput contiguous
 * (gob, D*D, oh, ow), both gradients, no workspace. */
int dtt_correlation_backward_nhwc(const float* gradOutput, int gob, int goc, int goh, int gow, const float* input1, int ic,
                                  int ih, int iw, const float* input2, float* gradInput1, float* gradInput2, int pad_size,
                                  int kernel_size, int max_displacement, int stride1, int stride2, void* stream);
int dtt_correlation_backward_nhwc_strided(const float* gradOutput, long g_batch_stride, long g_ch_stride, long g_px_stride, int gob,
                                          int goc, int goh, int gow, const float* input1, int ic, int ih, int iw, const float* input2,
                                          float* gradInput1, float* gradInput2, int pad_size, int kernel_size, int max_displacement,
                                          int stride1, int stride2, int which, void* workspace, size_t workspace_bytes, void* stream);
/* The same op in two phases (round 6): phase 1 = lay out the band words in the workspace (reads gradOutput only; event tag
 * "corr_bwd_band"), phase 2 = the gradients from a workspace that a phase-1 call with the same arguments filled ("corr_bwd_op"),
 * 3 = both (= dtt_correlation_backward_nhwc_strided).  The phases may be issued on different streams, ordered by the caller
 * (dtt.heads.TrackingRowsFn with DTT_CORR_BWD_OVERLAP=1: the bands of conv4 / conv5 beside conv3's gradient op -- measured: the ops
 * get shorter, the step longer; off by default). */
int dtt_correlation_backward_nhwc_phase(const float* gradOutput, long g_batch_stride, long g_ch_stride, long g_px_stride, int gob,
                                        int goc, int goh, int gow, const float* input1, int ic, int ih, int iw, const float* input2,
                                        float* gradInput1, float* gradInput2, int pad_size, int kernel_size, int max_displacement,
                                        int stride1, int stride2, int which, int phase, void* workspace, size_t workspace_bytes,
                                        void* stream);
size_t dtt_correlation_backward_workspace_bytes(int batch, int ic, int ih, int iw, int pad_size, int kernel_size,
                                                int max_displacement, int stride1, int stride2);
/* 1 if the streamed gradient kernels cover the geometry (kernel_size 1, stride1 == stride2 | max_displacement, radius <= 8,
 * ic % 64 == 0). */
int dtt_correlation_backward_stream_supported(int ic, int kernel_size, int max_displacement, int stride1, int stride2);
/* Test hooks (pure host code).  _plan_check replays every work item of the plan the launcher would use for `batch` images of
 * target_h x target_w target pixels on `compute_units` CUs through the kernel's own item decode: 1 iff every (image, 4 x 4 target
 * block, 64-channel group) is owned by exactly one wave, the dispatch order is a permutation of the items and every tile shape fits
 * its LDS ring.  _plan reports the plan (work items, channel groups per item, dynamic LDS bytes, whether the order rides in the
 * kernel arguments). */
int dtt_correlation_backward_plan_check(int batch, int target_h, int target_w, int window_radius, int channels, int compute_units);
int dtt_correlation_backward_plan(int batch, int target_h, int target_w, int window_radius, int channels, int compute_units,
                                  int* items, int* chunk, int* lds_bytes, int* table);

/* ---------------------------------------------------------------- PSRoI pooling
 * Replaces PSROIPoolForwardLauncher  (psroi_pooling/src/psroi_pooling_kernel.cu:82-106)
 *      and PSROIPoolBackwardLauncher (psroi_pooling/src/psroi_pooling_kernel.cu:172-194).
 * batch_size is an extra argument (the reference forward never needed it).  mapping_channel may
 * be NULL (the channel is a pure function of the output index).  The backward writes every
 * element of bottom_diff (no pre-zeroing needed) and uses LDS atomics only.
 */
int dtt_psroi_pool_forward(const float* bottom_data, float spatial_scale, int batch_size,
                           int num_rois, int height, int width, int channels,
                           int pooled_height, int pooled_width, const float* bottom_rois,
                           int group_size, int output_dim, float* top_data,
                           int* mapping_channel, void* stream);
int dtt_psroi_pool_backward(const float* top_diff, const int* mapping_channel, int batch_size,
                            int num_rois, float spatial_scale, int channels, int height,
                            int width, int pooled_width, int pooled_height, int output_dim,
                            int group_size, float* bottom_diff, const float* bottom_rois,
                            void* stream);
/* Fused PSRoI pool + the 7x7 average vote that always follows it (rfcn.py:62-64,136-140,196):
 * vote_out is (num_rois, output_dim); top_data may be NULL. */
int dtt_psroi_pool_vote_forward(const float* bottom_data, float spatial_scale, int batch_size,
                                int num_rois, int height, int width, int channels,
                                int pooled_height, int pooled_width, const float* bottom_rois,
                                int group_size, int output_dim, float* top_data,
                                float* vote_out, void* stream);
/* Vote only (what the R-FCN heads consume at inference): the pooled bins never take the reference layout -- they go
 * to `scratch` (channels * num_rois floats, channel-major: coalesced stores and coalesced vote reads) and only
 * vote_out (num_rois, output_dim) is defined on return.  Same bin arithmetic, summation order and division as the
 * two-step path, so vote_out is bit-identical to dtt_psroi_pool_vote_forward's. */
int dtt_psroi_vote_forward(const float* bottom_data, float spatial_scale, int batch_size, int num_rois, int height,
                           int width, int channels, int pooled_height, int pooled_width, const float* bottom_rois,
                           int group_size, int output_dim, float* scratch, float* vote_out, void* stream);

/* ---------------------------------------------------------------- NMS
 * Replaces nms_cuda_compute (nms/src/nms_cuda_kernel.cu:87-161).  boxes: (boxes_num, boxes_dim>=4)
 * rows [x1,y1,x2,y2,...] already sorted by descending score.  keep_out: int32[boxes_num],
 * num_out: int32[1].  Device in -> device out, stream ordered: the greedy sweep
 * (nms_cuda_kernel.cu:131-144) runs on the GPU, nothing is copied to the host.
 * max_keep > 0 lets the sweep stop after max_keep survivors (the proposal layer keeps only the
 * first post_nms_topN, proposal_layer.py:151-152); 0 = keep all (reference behaviour).  With max_keep << boxes_num the
 * IoU bit matrix is computed in two stream-ordered phases -- the rows the sweep needs first, the rest only if the keep list
 * is still short -- with the same keep list (the workspace holds the matrix, the parked sweep state, and one word per box:
 * the overlaps with the earlier boxes of its own 64-box chunk, from which the sweep settles a chunk in a few wave-wide steps).
 * boxes_num <= 65408: the sweep keeps one removal word per 64 boxes next to its 156 KB of staging area in one CU's LDS (the
 * call fails with a message beyond that; the reference has no such bound but needs boxes_num^2 / 8 bytes of mask either way).
 */
size_t dtt_nms_workspace_bytes(int boxes_num);
int dtt_nms(int* keep_out, int* num_out, const float* boxes, int boxes_num, int boxes_dim,
            float nms_overlap_thresh, int max_keep, void* workspace, size_t workspace_bytes,
            void* stream);

/* ---------------------------------------------------------------- RoI Align
 * Replaces ROIAlignForwardLaucher  (roi_align/src/roi_align_kernel.cu:73-91)
 *      and ROIAlignBackwardLaucher (roi_align/src/roi_align_kernel.cu:145-162).
 * pool_mode: 0 = raw aligned_height x aligned_width samples (RoIAlign, modules/roi_align.py:6-16);
 *            1 = samples on (h+1)x(w+1) then 2x2/s1 average (RoIAlignAvg, :18-29), fused;
 *            2 = same with max (RoIAlignMax, :31-42), fused.
 * For modes 1/2 aligned_height/width are the OUTPUT sizes (e.g. 7).  bottom_diff must be zeroed
 * by the caller (the backward accumulates with atomics, as the reference does).
 */
int dtt_roi_align_forward(const float* bottom_data, float spatial_scale, int num_rois, int height,
                          int width, int channels, int aligned_height, int aligned_width,
                          const float* bottom_rois, float* top_data, int pool_mode, void* stream);
/* Same result (bit-identical), map-stationary: needs the batch size (the reference launcher does not carry it), keeps
 * kAlignCB channel planes of one image in LDS and computes each output bin's geometry once for all of them.  Falls
 * back to the kernel above when the planes do not fit LDS.  RoIs with a batch index outside [0, batch_size) are skipped. */
int dtt_roi_align_forward_planes(const float* bottom_data, float spatial_scale, int batch_size, int num_rois,
                                 int height, int width, int channels, int aligned_height, int aligned_width,
                                 const float* bottom_rois, float* top_data, int pool_mode, void* stream);
int dtt_roi_align_backward(const float* top_diff, float spatial_scale, int batch_size,
                           int num_rois, int height, int width, int channels, int aligned_height,
                           int aligned_width, const float* bottom_rois, float* bottom_diff,
                           void* stream);

/* ---------------------------------------------------------------- RoI (max) pooling
 * Replaces ROIPoolForwardLaucher  (roi_pooling/src/roi_pooling_kernel.cu:95-126)
 *      and ROIPoolBackwardLaucher (roi_pooling/src/roi_pooling_kernel.cu:205-234).
 * bottom_diff must be zeroed by the caller: the backward scatters top_diff through argmax after
 * re-applying the admission tests of the reference's O(pixels x RoIs) gather (same result).
 */
int dtt_roi_pool_forward(const float* bottom_data, float spatial_scale, int num_rois, int height,
                         int width, int channels, int pooled_height, int pooled_width,
                         const float* bottom_rois, float* top_data, int* argmax_data,
                         void* stream);
int dtt_roi_pool_backward(const float* top_diff, float spatial_scale, int batch_size, int num_rois,
                          int height, int width, int channels, int pooled_height,
                          int pooled_width, const float* bottom_rois, float* bottom_diff,
                          const int* argmax_data, void* stream);

/* ---------------------------------------------------------------- RoI crop (bilinear grid sampler)
 * Replaces BilinearSamplerBHWD_updateOutput_cuda_kernel    (roi_crop/src/roi_crop_cuda_kernel.cu:201-255)
 *      and BilinearSamplerBHWD_updateGradInput_cuda_kernel (roi_crop/src/roi_crop_cuda_kernel.cu:257-326).
 * inputImages (ib, ic, ih, iw) NCHW; grids (ob, oh, ow, 2) holding (y, x) in [-1,1]; output
 * (ob, ic, oh, ow); RoI b samples image b / (ob / ib).  Output is fully written (samples falling
 * outside give 0).  gradInputImages must be zeroed by the caller; as in the reference the grid
 * gradient is not produced (roi_crop_cuda_kernel.cu:154-192 computes and drops it).
 */
int dtt_roi_crop_forward(int oc, int ow, int oh, int ob, int ic, int ih, int iw, int ib,
                         const float* inputImages, const float* grids, float* output,
                         void* stream);
int dtt_roi_crop_backward(int goc, int gow, int goh, int gob, int ic, int ih, int iw, int ib,
                          const float* inputImages, const float* grids, float* gradInputImages,
                          const float* gradOutput, void* stream);

/* ---------------------------------------------------------------- RPN proposal layer
 * Replaces _ProposalLayer.forward (rpn/proposal_layer.py:49-161) = anchors + bbox_transform_inv
 * (rpn/bbox_transform.py:108-134) + clip_boxes (:156-173) + torch.sort + per-image nms loop.
 * One call handles the whole batch on the device with no host round trip.
 *   cls_prob (B, 2A, H, W)   fg probability of anchor a at channel A + a (proposal_layer.py:67)
 *   bbox_pred (B, 4A, H, W)  delta of anchor a at channels 4a..4a+3 (proposal_layer.py:98-99)
 *   im_info (B, 3) = [height, width, scale];  anchors (A, 4) base anchors (generate_anchors.py:45-56)
 *   rois_out (B, post_nms_topN, 5): [batch index, x1, y1, x2, y2], zero padded (proposal_layer.py:157-159)
 *   num_out  int32[B] number of valid rows per image (may be NULL)
 * Ordering: scores descending, ties broken by lower anchor index first (torch.sort on CUDA 0.3 was
 * unspecified; this is the declared order, SURVEY.md section 7 hard part 3).
 * pre_nms_topN follows the reference guard (proposal_layer.py:138-139): it is applied only when
 * 0 < pre_nms_topN < B*K*A.
 */
size_t dtt_proposal_workspace_bytes(int batch, int num_anchors, int height, int width,
                                    int pre_nms_topN);
int dtt_proposal_forward(const float* cls_prob, const float* bbox_pred, const float* im_info,
                         const float* anchors, int batch, int num_anchors, int height, int width,
                         int feat_stride, int pre_nms_topN, int post_nms_topN, float nms_thresh,
                         float* rois_out, int* num_out, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The same layer in two stream-ordered phases, for callers that overlap it with other work (dtt/model.py starts phase 1 on
 * a side stream right after the RPN softmax, while the box-delta convolution still runs on the main stream):
 *   dtt_proposal_select_sort  reads the scores only: per image, the pre_nms_topN best (score, anchor) keys in order -> workspace
 *   dtt_proposal_decode_nms   decodes + clips those anchors' boxes (bbox_pred, im_info, anchors), NMS, writes rois_out / num_out
 * Both take the geometry arguments and the workspace of dtt_proposal_forward; the workspace carries the selection from one to
 * the other.  dtt_proposal_forward gives the same RoIs with fewer launches: having the box deltas at hand, the kernel that
 * ranks the selected anchors also decodes their boxes.
 * The selection runs on many workgroups (runs of 1024 anchors sorted in LDS, then ranked against each other: exact
 * (score desc, anchor index asc) order, no atomics) when K*A <= 38912; larger maps take one workgroup per image. */
int dtt_proposal_select_sort(const float* cls_prob, int batch, int num_anchors, int height, int width, int pre_nms_topN,
                             void* workspace, size_t workspace_bytes, void* stream);
int dtt_proposal_decode_nms(const float* bbox_pred, const float* im_info, const float* anchors, int batch, int num_anchors,
                            int height, int width, int feat_stride, int pre_nms_topN, int post_nms_topN, float nms_thresh,
                            float* rois_out, int* num_out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- RPN anchor-target layer
 * Replaces _AnchorTargetLayer.forward (rpn/anchor_target_layer.py:48-191) with bbox_overlaps_batch
 * (rpn/bbox_transform.py:208-254) and bbox_transform_batch (:36-75).  Stream-ordered phases, so that
 * the random subsampling can consume numpy's RNG on the host exactly as the reference does
 * (anchor_target_layer.py:124-141):
 *   dtt_anchor_target_assign : per-anchor labels {1,0,-1} before subsampling (-1 also for anchors
 *     outside the image), the arg-max GT per anchor, and per-image [fg, bg] counts.
 *   host: reads the 2B counts, draws the permutations, uploads the indices to disable.
 *   dtt_anchor_target_disable: labels[b, disable[i]] = -1 for i in [offsets[b], offsets[b+1]).
 *   dtt_anchor_target_finish : regression targets, weights, and the four outputs in the reference
 *     layouts (anchor_target_layer.py:168-189): labels (B,1,A*H,W), targets / inside / outside
 *     weights (B,4A,H,W).
 * gt_boxes (B, G, 5) [x1,y1,x2,y2,cls].  im_h0 / im_w0 = long(im_info[0][0]), long(im_info[0][1]):
 * the reference tests "inside the image" against image 0 only (anchor_target_layer.py:85-86,
 * reproduced); passing them by value keeps the call free of device-to-host copies.
 * positive_weight / negative_weight are the outside weights of fg / bg anchors
 * (1/num_examples of the LAST image under the reference's uniform weighting, :153-156).
 * gt_max_scratch: int32[B*G] scratch.
 */
int dtt_anchor_target_assign(const float* gt_boxes, int im_h0, int im_w0, const float* anchors,
                             int batch, int num_gt, int num_anchors, int height, int width,
                             int feat_stride, float negative_overlap, float positive_overlap,
                             int clobber_positives, int* labels /* (B, K*A) */,
                             int* argmax_gt /* (B, K*A) */, int* counts /* (B, 2) */,
                             int* gt_max_scratch, void* stream);
int dtt_anchor_target_disable(int* labels, const int* disable, const int* disable_offsets,
                              int batch, int total_anchors, void* stream);
int dtt_anchor_target_finish(const float* gt_boxes, int im_h0, int im_w0, const float* anchors,
                             const int* labels_in, const int* argmax_gt, int batch, int num_gt,
                             int num_anchors, int height, int width, int feat_stride,
                             float inside_weight, float positive_weight, float negative_weight,
                             float* labels_out, float* bbox_targets, float* bbox_inside_weights,
                             float* bbox_outside_weights, void* stream);
/* The same layer without a host read (cfg.TRAIN.SAMPLER_RNG = "device", the counterpart of the RoI sampler's device mode): the
 * three phases in one call.  im_info (B, 3) stays on the device (row 0 is read by the kernels); instead of numpy permutations
 * of the index lists (anchor_target_layer.py:124-141) every anchor carries a 32-bit random key the caller drew WITHOUT looking at
 * the labels, and of a class that exceeds its quota the candidates with the smallest (key, anchor index) stay -- a uniform subset
 * of exactly the reference's size: num_fg foreground anchors, rpn_batchsize - (foreground count before subsampling) background
 * anchors.  The outside weights (1 / num_examples of the LAST image, or positive_weight / n_pos and (1 - positive_weight) / n_neg
 * when positive_weight >= 0) are computed on the device.  keys: (B, K*A) uint32.  Scratch: labels / argmax_gt (B, K*A) int32,
 * counts (B, 4) int32 (per-image [fg, bg] before, then after subsampling), gt_max_scratch (B*G) int32, weights (2) float.
 */
int dtt_anchor_target_device(const float* gt_boxes, const float* im_info, const float* anchors, const unsigned* keys,
                             int batch, int num_gt, int num_anchors, int height, int width, int feat_stride,
                             int rpn_batchsize, int num_fg, float negative_overlap, float positive_overlap,
                             int clobber_positives, float inside_weight, float positive_weight, int* labels,
                             int* argmax_gt, int* counts, int* gt_max_scratch, float* weights, float* labels_out,
                             float* bbox_targets, float* bbox_inside_weights, float* bbox_outside_weights, void* stream);

/* ---------------------------------------------------------------- RoI / tracking target samplers (training)
 * Replace _ProposalTargetLayer._sample_rois_pytorch (rpn/proposal_target_layer_cascade.py:121-208: IoU against the
 * ground truth, fg / bg classification, random sampling of rois_per_image RoIs, target encoding + normalisation; its
 * per-image / per-RoI Python loops and host syncs) and _TrackingProposalTargetLayer.forward
 * (rpn/tracking_proposal_target_layer.py:33-196).
 * Candidates of an image = its num_rois proposals (all_rois (images, num_rois, 5)) followed by its num_gt ground-truth
 * boxes (gt_boxes (images, num_gt, gt_stride >= 5) rows [x1,y1,x2,y2,cls,..]; num_gt <= 64), as :42-46 appends them.
 *   assign   int32 (images, num_rois + num_gt): arg-max ground-truth row of every candidate (first maximum)
 *   fg_list / bg_list  int32, same shape: candidate indices with max IoU >= fg_thresh / in [bg_thresh_lo, bg_thresh_hi),
 *            ascending (what torch.nonzero returns in the reference); counts int32 (images, 2) = their lengths
 * dtt_proposal_target_sample then writes rois (images, n_out, 5), labels (images, n_out), targets / inside / outside
 * (images, n_out, 4) and status int32 (images) (1 = no candidate of either kind: the reference raises there).  Which
 * candidates: EITHER pos int32 (images, n_out) + fg_n int32 (images) (device arrays: slot j < fg_n[b] takes
 * fg_list[b][pos[b][j]], the others bg_list[b][pos[b][j]] -- the host has drawn them from its generator after reading
 * counts) OR, with both NULL, u_fg (images, num_rois + num_gt) and u_bg (images, n_out) float64 uniforms in [0, 1) drawn
 * without knowledge of the counts: the fg_n = min(fg_per_image, fg count) smallest keys u_fg[b][k], k < fg count, select a
 * uniform foreground subset without replacement, background slot j takes bg_list[b][floor(u_bg[b][j - fg_n] * bg count)]
 * (with only one kind of candidate, all n_out slots index it through u_bg, as :165-181).  mean4 / std4 / inside4 are HOST
 * arrays of 4 floats (cfg.TRAIN.BBOX_NORMALIZE_MEANS / _STDS / BBOX_INSIDE_WEIGHTS).
 * dtt_tracking_target: gt_boxes (2, images, num_gt, 6) [x1,y1,x2,y2,cls,track_id], num_boxes int64 (2, images); outputs as
 * above with n_out = num_gt. */
int dtt_proposal_target_assign(const float* all_rois, const float* gt_boxes, int images, int num_rois, int num_gt,
                               int gt_stride, float fg_thresh, float bg_thresh_hi, float bg_thresh_lo, int* assign,
                               int* fg_list, int* bg_list, int* counts, void* stream);
int dtt_proposal_target_sample(const float* all_rois, const float* gt_boxes, int images, int num_rois, int num_gt,
                               int gt_stride, const int* assign, const int* fg_list, const int* bg_list, const int* counts,
                               const int* pos, const int* fg_n, const double* u_fg, const double* u_bg, int n_out,
                               int fg_per_image, const float* mean4_host, const float* std4_host,
                               const float* inside4_host, int normalize, float* rois_out, float* labels_out,
                               float* targets_out, float* inside_out, float* outside_out, int* status, void* stream);
int dtt_tracking_target(const float* gt_boxes, const long* num_boxes, int images, int num_gt, const float* mean4_host,
                        const float* std4_host, const float* inside4_host, int normalize, float* rois_out,
                        float* labels_out, float* targets_out, float* inside_out, float* outside_out, void* stream);

/* ---------------------------------------------------------------- test-time per-class NMS
 * Replaces the per-class loop of the reference's test driver (test_net.py:274-301): for every class j >= 1
 * threshold scores[:, j] > score_thresh, sort descending (ties: lower RoI index first), NMS(nms_thresh), then the
 * max_per_image cut over all classes (keep score >= the max_per_image-th best) -- one call for all classes of
 * `images` images, nothing leaves the device.
 *   scores (images, num_rois, num_classes); boxes (images, num_rois, 4) if class_agnostic else (.., 4*num_classes)
 *   dets_out (images, num_classes, num_rois, 5) rows [x1,y1,x2,y2,score] in kept order (class 0 unused)
 *   count_out int32 (images, num_classes); num_rois <= 1024; max_per_image <= 0 disables the cut
 */
int dtt_class_nms(const float* scores, const float* boxes, int images, int num_rois, int num_classes,
                  int class_agnostic, float score_thresh, float nms_thresh, int max_per_image,
                  float* dets_out, int* count_out, void* stream);

/* ---------------------------------------------------------------- fused trunk epilogue
 * No custom-op counterpart in the reference: replaces the separate BatchNorm (frozen, faster_rcnn/resnet.py:
 * 290-295, 325-330) / residual add / ReLU passes of its ResNet blocks (resnet.py:88-107) once the BatchNorm
 * affine has been folded into the convolution.  In place: x[n,c,:] = act(x[n,c,:] + bias[c] (+ residual[n,c,:])).
 * x (batch, channels, hw) contiguous fp32; residual may be NULL; batch*channels <= 65535 per call.
 */
int dtt_bias_act_inplace(float* x, const float* bias, const float* residual, int batch, int channels,
                         int hw, int relu, void* stream);
/* Same epilogue for the channels-last trunk: x (rows, channels) row-major, bias along the fastest dimension.
 * channels % 4 == 0; x, bias and residual 16-byte aligned; residual may be NULL. */
int dtt_bias_act_nhwc_inplace(float* x, const float* bias, const float* residual, long rows, int channels,
                              int relu, void* stream);

/* Strided-batched row-major GEMM: out[b] (rows, n) = a[b] (rows, k) * w[b] (k, n) for b < batch, operands packed back to
 * back.  Library GEMM (hipBLASLt, candidates timed once per shape); no reference counterpart -- it carries the 16 / 36 products
 * of the Winograd form of the trunk's 3x3 convolutions (faster_rcnn/resnet.py:76-78 `conv3x3`). */
int dtt_gemm_batched(float* out, const float* a, const float* w, int batch, long rows, int k, int n, void* workspace,
                     size_t workspace_bytes, void* stream);
/* One-time candidate timing for a shape of dtt_gemm_batched: launches the library's candidates on the caller's operands
 * (`out` receives the product) and SYNCHRONISES the stream; the winner is remembered per (device, shape).  Call it before
 * the first dtt_gemm_batched of a shape, outside timed / captured regions.  Without it the first heuristic runs. */
int dtt_gemm_batched_tune(float* out, const float* a, const float* w, int batch, long rows, int k, int n, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Winograd transforms for a 3x3, stride-1, padding == dilation convolution on channels-last maps (the reference's
 * faster_rcnn/resnet.py:76-78, 307 are plain nn.Conv2d calls).  m = 2: F(2x2, 3x3), 16 products; m = 4: F(4x4, 3x3), 36
 * products.  tiles = dtt_winograd_tiles(images, height, width, dilation, m); t2 = (m + 2)^2.
 * Input: x (images, height, width, channels) -> v[t2][tiles][channels].  Output: mm[t2][tiles][channels] -> y (images,
 * height, width, channels) = A^T mm A + bias[c], ReLU if relu != 0.  channels % 4 == 0.  The products mm[i] = v[i] * u[i]
 * in between are dtt_gemm_batched's. */
long dtt_winograd_tiles(int images, int height, int width, int dilation, int m);
int dtt_winograd_input_transform(const float* x, float* v, int images, int height, int width, int channels, int dilation,
                                 int m, void* stream);
int dtt_winograd_output_transform(const float* mm, const float* bias, float* y, int images, int height, int width,
                                  int channels, int dilation, int m, int relu, void* stream);

/* Batched 2-D transpose: in (batch, rows, cols) row-major -> out (batch, cols, rows).  The layout change between
 * the channels-last trunk and the NCHW maps the operators above read (NHWC -> NCHW: rows = H*W, cols = C), which the
 * reference never needs because its trunk is NCHW throughout (faster_rcnn/resnet.py:325-343).  in != out. */
int dtt_transpose_batched(const float* in, float* out, int batch, int rows, int cols, void* stream);
/* n independent row scalings in as few launches as the kernel-argument space allows (48 tensors each):
 * dst[i][r][c] = src[i][r][c] * scale[i][r], tensor i being rows[i] x cols[i], dense, row-major in memory order (a (K, C, kh, kw)
 * convolution filter in either NCHW or channels-last memory is K rows of C*kh*kw).  src / scale / dst / rows / cols are HOST
 * arrays of n entries holding device pointers / extents.  The training trunk folds the frozen BatchNorm scales into ~100
 * filters per step with it (dtt/fuse.py: _FoldScalesFn, forward and backward), instead of ~100 launch-bound multiplies. */
int dtt_scale_rows_batch(int n, const float* const* src, const float* const* scale, float* const* dst, const int* rows,
                         const int* cols, void* stream);

/* Column blocks of a row-major matrix gathered side by side: dst[r][k * ncols + c] = src[k * src_block_rows + r][c] for
 * k < n_blocks, r < rows, c < ncols (dst / src point at the first column of interest; leading dimensions in floats).
 * This is rfcn.py:133-140's `torch.cat((bbox_t, bbox_t+tau, ...), dim=1)` for the two legs' box-delta columns on
 * position-major rows (the correlation kernels write their columns of the same rows directly).  16-byte aligned pointers,
 * ncols / dst_ld / src_ld multiples of 4. */
int dtt_gather_column_blocks(float* dst, long dst_ld, const float* src, long src_ld, long src_block_rows, int n_blocks,
                             long rows, int ncols, void* stream);

/* ResNet stem tail on a channels-last map whose frozen-BatchNorm scale is folded into conv1 (faster_rcnn/resnet.py:110-117:
 * bn1 -> relu -> MaxPool2d(kernel 3, stride 2, padding 0, ceil_mode=True)): y = relu(maxpool(x) + bias[c]) in one pass --
 * the same values as maxpool(relu(x + bias)), since adding a per-channel constant and clamping at zero are monotonic.
 * x (images, height, width, channels) -> y (images, oh, ow, channels), oh = ceil((height - 3) / 2) + 1 (same for ow);
 * channels % 4 == 0, 16-byte aligned pointers, x != y. */
int dtt_maxpool3s2_bias_relu_nhwc(const float* x, const float* bias, float* y, int images, int height, int width,
                                  int channels, void* stream);

/* Row-major GEMM with the bottleneck epilogue: out (rows, n) = act(a (rows, k) * w (k, n) + bias[n] (+ residual
 * (rows, n))); residual may be NULL and may alias out.  A library GEMM (hipBLASLt) -- the entry point exists for the
 * epilogue: frozen-BatchNorm shift + `out += residual` + ReLU of faster_rcnn/resnet.py:100-107 in the GEMM itself.
 * workspace: caller-owned scratch for the library (32 MiB is plenty); plans are cached per (device, shape). */
int dtt_gemm_bias_act(float* out, const float* a, const float* w, const float* bias, const float* residual,
                      long rows, int k, int n, int relu, void* workspace, size_t workspace_bytes, void* stream);
/* One-time candidate timing for a shape (and epilogue variant) of dtt_gemm_bias_act: the products go to `scratch` (rows * n
 * floats owned by the caller), a / w / bias / residual are only read; launches every candidate a few times and
 * SYNCHRONISES the stream -- the only entry points of this library that do (with dtt_gemm_batched_tune); the hot entry point
 * above never allocates, frees or synchronises and runs the first heuristic for a shape that was never tuned. */
int dtt_gemm_tune(const float* a, const float* w, const float* bias, const float* residual, long rows, int k, int n,
                  int relu, float* scratch, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- R-FCN 1x1 heads + position-major PSRoI pooling
 * dtt_head_gemm replaces the cuDNN 1x1 convolutions RFCN_cls_net / RFCN_bbox_net / corr_bbox_net
 * (faster_rcnn/rfcn.py:49-53, 133-134, 194; resnet.py:311-312) at inference: an exact-fp32 MFMA GEMM over channels-last
 * pixel rows.   out[m][n] = sum_k x[m][k] * w[n][k] + bias[n]  for n < n_store.
 *   x (M, K) rows ldx floats apart, K % 32 == 0;  w (n_rows, K) row-major, n_rows % 16 == 0, rows in the order the
 *   channels are to be emitted (zero rows as padding);  bias (n_rows);  out (M, ldc);  n_store % 4 == 0;  all pointers
 *   16-byte aligned.  passes = 0 lets the library choose how many sub-ranges of the channel tiles a workgroup walks
 *   (stores of one sub-range drain under the MFMAs of the next).
 * The callers emit the position-major layout  n = head_offset + bin*cp + ctop  (bin = ph*G + pw; reference channel
 * (ctop*G + ph)*G + pw, psroi_pooling_kernel.cu:62), which dtt_psroi_pm_forward pools:
 * PSROIPoolForward (psroi_pooling_kernel.cu:15-79) + the G x G average vote (rfcn.py:62-64, 136-140) with lanes = classes.
 *   map: first float of the head inside pixel 0; pixel (b, h, w) starts pixel_stride * ((b*height + h)*width + w) floats
 *   later; cp = classes-per-bin padding (4 or 32), output_dim <= cp; rois (num_rois, 5);
 *   vote_out (num_rois, output_dim); pooled_out (num_rois, output_dim, pooled, pooled) or NULL.
 * Same bin arithmetic and summation order as dtt_psroi_pool_vote_forward on the equivalent NCHW map: bit-identical. */
int dtt_head_gemm(const float* x, long ldx, int M, int K, const float* w, const float* bias, int n_rows,
                  float* out, long ldc, int n_store, int passes, void* stream);
/* The RPN's two 1x1 heads + the pairwise softmax in ONE launch of the same kernel (rpn/rpn.py:63-71: RPN_cls_score ->
 * reshape(2) -> softmax -> reshape(2A), RPN_bbox_pred).  x: (batch * hw, K) channels-last rows of relu(RPN_Conv(.)), ldx floats
 * between rows.  w: (n_rows, K) with the rows in the order [bg_0, fg_0, bg_1, fg_1, ..., bg_{A-1}, fg_{A-1}, box deltas
 * 0 .. 4A-1, zero rows up to a multiple of 16] where bg_a / fg_a are RPN_cls_score's output channels a / A + a (the pair the
 * reference's softmax normalises), bias likewise.  cls_prob (batch, 2A, h, w) and bbox_pred (batch, 4A, h, w) come out in the
 * reference's NCHW layout -- what dtt_proposal_select_sort / dtt_proposal_decode_nms read.  num_anchors must be even. */
int dtt_rpn_head_gemm(const float* x, long ldx, int batch, int hw, int K, const float* w, const float* bias, int n_rows,
                      int num_anchors, float* cls_prob, float* bbox_pred, void* stream);

/* Backward of dtt_rpn_head_gemm's epilogue (training graph, rpn.py:63-71): the gradients with respect to cls_prob (batch, 2A, h, w)
 * and bbox_pred (batch, 4A, h, w) -- either may be NULL = zero -- become the (batch * hw, ld) rows of the packed GEMM's output
 * gradient, columns [bg_0, fg_0, ..., bg_{A-1}, fg_{A-1}, box deltas 0 .. 4A-1, zeros up to ld], with the adjoint of the pairwise
 * softmax applied (cls_prob: the forward's probabilities).  dX and dW then are dtt_head_gemm / dtt_head_gemm_dw over these rows. */
int dtt_rpn_head_grad_rows(const float* grad_cls_prob, const float* grad_bbox_pred, const float* cls_prob, int batch, int hw,
                           int num_anchors, float* rows, long ld, int cls_grad_is_logits, void* stream);
/* (cls_grad_is_logits = 1: grad_cls_prob already is the gradient with respect to the score LOGITS -- dtt_rpn_loss_backward's --
 *  and is copied into its columns as it is; 0: the gradient with respect to the probabilities, softmax adjoint applied.) */

/* The two RPN losses of rpn/rpn.py:86-105 for `legs` legs (frames of a pair) of batch / legs images each, one launch:
 * loss[leg] = cross-entropy over the anchors the anchor-target layer labelled 0 / 1 (labels: (batch, 1, A*h, w) floats, -1 = not
 * sampled; rpn.py:90-97 gathers them with nonzero() + index_select -- the mean over the labelled anchors is the same number) read
 * as -log cls_prob[label] of the pairwise-softmaxed scores (dtt_rpn_head_gemm's cls_prob); loss[legs + leg] = _smooth_l1_loss(
 * bbox_pred, bbox_targets, inside, outside, sigma, dim = [1, 2, 3]) (net_utils.py:73-87, rpn.py:104-105).  count[leg]: the leg's
 * labelled anchors.  Deterministic (per-workgroup partials added in index order by the last workgroup).  workspace:
 * dtt_rpn_loss_workspace_bytes(batch, hw) bytes, caller-owned. */
size_t dtt_rpn_loss_workspace_bytes(int batch, int hw);
int dtt_rpn_loss_forward(const float* cls_prob, const float* bbox_pred, const float* labels, const float* bbox_targets,
                         const float* inside_weights, const float* outside_weights, int batch, int legs, int num_anchors, int hw,
                         float sigma, float* loss, float* count, void* workspace, size_t workspace_bytes, void* stream);
/* Gradient of sum_leg grad_loss[leg] * class loss + grad_loss[legs + leg] * box loss (grad_loss, count: device arrays):
 * grad_logits (batch, 2A, h, w) with respect to the score LOGITS, (p - y) * g / count -- what cross_entropy on the logits gives
 * (rpn.py:97), alive where a probability underflows -- and grad_bbox (batch, 4A, h, w) with respect to bbox_pred; every element
 * is written.  Feed both to dtt_rpn_head_grad_rows with cls_grad_is_logits = 1. */
int dtt_rpn_loss_backward(const float* cls_prob, const float* bbox_pred, const float* labels, const float* bbox_targets,
                          const float* inside_weights, const float* outside_weights, const float* grad_loss, const float* count,
                          int batch, int legs, int num_anchors, int hw, float sigma, float* grad_logits, float* grad_bbox, void* stream);

/* Weight gradient of the packed 1x1 heads (training graph, rfcn.py:49-53): dw[n][k] = sum_m gout[m][n] * x[m][k] for n < N,
 * k < K over the M pixel rows of the position-major maps -- gout (M, g_cols >= N columns readable, row stride ldg floats; the
 * columns N .. g_cols only have to be finite), x (M, K) with row stride ldx, dw (N, K) dense.  Exact-f32 MFMA; both operands
 * are read as they lie (no transposes); the pixel rows are split over workgroups whose partial tiles meet in `workspace`
 * (dtt_head_gemm_dw_workspace_bytes) and are added in a fixed order: deterministic, no atomics.  K, g_cols, ldg, ldx % 4 == 0,
 * 16-byte aligned pointers. */
size_t dtt_head_gemm_dw_workspace_bytes(int M, int N, int K);
int dtt_head_gemm_dw(const float* gout, long ldg, int g_cols, const float* x, long ldx, int M, int N, int K, float* dw,
                     void* workspace, size_t workspace_bytes, void* stream);
int dtt_psroi_pm_forward(const float* map, long pixel_stride, int cp, int batch_size, int num_rois, int height,
                         int width, int pooled, const float* rois, float spatial_scale, int output_dim,
                         float* vote_out, float* pooled_out, void* stream);
/* Detection pooling of every RoI in ONE launch (rfcn.py:133-140 at inference): class scores and box deltas of the same position-major
 * map pooled by one workgroup per RoI (same bin edges, same rows) and the softmax over the classes folded into the epilogue.  The class
 * head starts at float 0 of a pixel (32 slots per bin, n_cls <= 32 used), the box head at float loc_offset (4 per bin).  Votes are the
 * very sums of dtt_psroi_pm_forward.  cls_prob (num_rois, n_cls), loc_vote (num_rois, n_loc); cls_vote (num_rois, n_cls) or NULL. */
int dtt_psroi_pm_det_forward(const float* map, long pixel_stride, int loc_offset, int batch_size, int num_rois, int height, int width,
                             int pooled, const float* rois, float spatial_scale, int n_cls, int n_loc, float* cls_vote, float* cls_prob,
                             float* loc_vote, void* stream);
/* Backward of the vote with respect to the position-major map (PSROIPoolBackward, psroi_pooling_kernel.cu:109-170, composed with
 * the AvgPool2d of rfcn.py:62-64): grad_map[pixel][bin*cp + c] = sum over the RoIs of that image whose bin contains the pixel of
 * grad_vote[roi][c] / pooled^2 / bin_area, added in RoI order -- map-stationary, no atomics (the reference scatters with
 * atomicAdd), deterministic.  Every pixel's floats [0, pooled^2 * cp) are WRITTEN (zeros where no RoI reaches): no pre-zeroing.
 * edges: caller-owned scratch, num_rois * (4 * pooled + 1) + 2 * batch_size ints (the bin edges of every RoI, then the run of RoI
 * indices of every image: a pixel's wave only walks the RoIs of its own image).  cp in {4, 32} as in the forward. */
int dtt_psroi_pm_backward(const float* grad_vote, const float* rois, int num_rois, int batch_size, int height, int width,
                          int pooled, float spatial_scale, int output_dim, int cp, long pixel_stride, float* grad_map,
                          int* edges, void* stream);
/* The same gradient for up to TWO heads pooled from one map with the same RoIs (the class and box heads of rfcn.py:133-150: the
 * reference runs PSROIPoolBackward once per head), in ONE launch: one wave per pixel lists the (RoI, bin) pairs that cover it once
 * and serves both heads.  Head 0's bins start at column 0 of a pixel's row (cp0 columns per bin), head 1's at pooled^2 * cp0
 * (cp1 per bin; grad_vote1 NULL = one head); cp0 + cp1 <= 64.  Columns [0, row_floats) of EVERY pixel are written -- the sums,
 * zeros where no RoI reaches and in the padding columns behind the heads -- so row_floats = pixel_stride hands the head GEMM's
 * backward a whole gradient map.  add_cols (pixels, add_count) or NULL: a compact gradient added into columns [add_first,
 * add_first + add_count) behind the pooling sums (the map's second consumer under autograd: the box deltas the tracking branch
 * concatenates, rfcn.py:166-169).  pooled = 7.  No scratch: the bin edges are computed inside the launch. */
int dtt_psroi_pm_backward_heads(const float* grad_vote0, int output_dim0, int cp0, const float* grad_vote1, int output_dim1, int cp1,
                                const float* rois, int num_rois, int batch_size, int height, int width, int pooled, float spatial_scale,
                                long pixel_stride, int row_floats, const float* add_cols, int add_first, int add_count,
                                float* grad_map, void* stream);

/* ---------------------------------------------------------------- zero-jump Viterbi tube linking
 * Replaces VideoPostProcessor._make_tubes / _zero_jump_link / _score_of_edge (lib/model/utils/tracking_utils.py:
 * 86-124, 127-264, 268-290) for `problems` independent (video, class) problems of `frames` frames each, in three
 * launches: per-frame NMS in the given priority order (first max_per_image <= 32 survivors), tracklet-link masks, and
 * all K = min_t n_t Viterbi paths per problem.  T = frames - 1 frames take part (the last one only closes the last
 * pair).  dets (problems, frames, max_dets <= 1024, det_stride >= 5) rows [x1,y1,x2,y2,score,...], n (problems,
 * frames) row counts; trk (problems, frames, 2, max_tracklets <= 512, 4) or NULL, m (problems, frames) tracklet
 * counts (-1 = the frame has none).  Outputs: kept_boxes (problems, T, 32, 4), kept_scores (problems, T, 32), kept_n
 * (problems, T), path_idx (problems, 32, T) box index per frame, path_total (problems, 32) = best score / frames,
 * n_paths (problems) -- 0 when a frame has no detection (the reference raises).  Ties: lowest index. */
size_t dtt_tube_link_workspace_bytes(int problems, int frames);
int dtt_tube_link(const float* dets, int det_stride, const int* n, const float* trk, const int* m, int problems,
                  int frames, int max_dets, int max_tracklets, int max_per_image, float nms_thresh, float* kept_boxes,
                  float* kept_scores, int* kept_n, int* path_idx, float* path_total, int* n_paths, void* workspace,
                  size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DTT_HIP_H */

// Training-time target samplers between the proposal layer and PSRoI pooling (gfx950).
//
// Replaces the per-image / per-RoI Python loops of _ProposalTargetLayer._sample_rois_pytorch
// (rpn/proposal_target_layer_cascade.py:121-208, with bbox_overlaps_batch bbox_transform.py:256-296 and
// bbox_transform_batch :54-70) and _TrackingProposalTargetLayer (rpn/tracking_proposal_target_layer.py:33-196):
//
//   pt_assign   one workgroup per image: every candidate (proposals + the ground-truth boxes appended to them, :42-46)
//               recomputes its <= 64 IoUs in registers (first maximum wins, as torch.max on the CPU), is classified
//               foreground / background by the reference's thresholds and compacted, in candidate order, into the two
//               index lists the reference builds with torch.nonzero -- ballots + popcounts, no atomics.
//   pt_sample   one workgroup per image: picks rois_per_image candidates out of those lists, gathers RoI / label /
//               matched box and encodes + normalises the regression targets.  Two ways to pick:
//                 * positions computed by the host from numpy's global RNG exactly as the reference draws them (needs the
//                   two counts on the host: one 8-byte read per image) -- bit-identical RoI batches for a given seed;
//                 * positions derived on the device from uniforms the host drew WITHOUT knowing the counts (a random key
//                   per candidate: the fg_n smallest keys of the foreground list are a uniform subset without
//                   replacement; background slots index floor(u * bg_count) with replacement, as the reference) -- the
//                   same distribution, nothing read back inside the training step.
//   tracking_target  one wave per image: track-id correspondence between the two frames, matched tracks sorted by id
//               and packed to the front (stable rank by counting), targets = encode(frame t box -> frame t+tau box).
#include "common.h"

namespace {

constexpr int kPtThreads = 1024;
constexpr int kMaxGt = 64;

// bbox_transform.py:256-296, 3-D branch: IoU with +1 extents; zero-area ground truth -> 0, zero-area candidate -> -1
__device__ __forceinline__ float iou_plus1(float ax1, float ay1, float ax2, float ay2, float gx1, float gy1, float gx2,
                                           float gy2) {
  const float gw = gx2 - gx1 + 1.f, gh = gy2 - gy1 + 1.f;
  const float aw = ax2 - ax1 + 1.f, ah = ay2 - ay1 + 1.f;
  const float g_area = gw * gh, a_area = aw * ah;
  float iw = fminf(ax2, gx2) - fmaxf(ax1, gx1) + 1.f;
  float ih = fminf(ay2, gy2) - fmaxf(ay1, gy1) + 1.f;
  iw = iw < 0.f ? 0.f : iw;
  ih = ih < 0.f ? 0.f : ih;
  const float inter = iw * ih;
  float ov = inter / (a_area + g_area - inter);
  if (gw == 1.f && gh == 1.f) ov = 0.f;
  if (aw == 1.f && ah == 1.f) ov = -1.f;
  return ov;
}

// grid = images.  all_rois (B, R, 5), gt (B, G, gt_stride >= 5) rows [x1,y1,x2,y2,cls,..].  Candidates 0..R-1 are the
// proposals, R..R+G-1 the ground-truth boxes.  Outputs per image: assign[N] (argmax gt), fg_list / bg_list [N]
// (candidate indices, ascending), counts[2].
__global__ __launch_bounds__(kPtThreads) void pt_assign(const float* __restrict__ all_rois, const float* __restrict__ gt,
                                                        int R, int G, int gt_stride, float fg_thresh, float bg_hi,
                                                        float bg_lo, int* __restrict__ assign, int* __restrict__ fg_list,
                                                        int* __restrict__ bg_list, int* __restrict__ counts) {
  __shared__ float sgt[kMaxGt * 4];
  __shared__ int wave_fg[kPtThreads / 64], wave_bg[kPtThreads / 64];
  __shared__ int base_fg, base_bg;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = R + G;
  for (int i = tid; i < G * 4; i += kPtThreads) sgt[i] = gt[((long)b * G + (i >> 2)) * gt_stride + (i & 3)];
  if (tid == 0) { base_fg = 0; base_bg = 0; }
  __syncthreads();
  for (int c0 = 0; c0 < N; c0 += kPtThreads) {
    const int c = c0 + tid;
    bool fg = false, bg = false;
    if (c < N) {
      float x1, y1, x2, y2;
      if (c < R) {
        const float* r = all_rois + ((long)b * R + c) * 5;
        x1 = r[1]; y1 = r[2]; x2 = r[3]; y2 = r[4];
      } else {
        x1 = sgt[(c - R) * 4]; y1 = sgt[(c - R) * 4 + 1]; x2 = sgt[(c - R) * 4 + 2]; y2 = sgt[(c - R) * 4 + 3];
      }
      float best = -INFINITY;
      int arg = 0;
      for (int g = 0; g < G; ++g) {
        const float ov = iou_plus1(x1, y1, x2, y2, sgt[g * 4], sgt[g * 4 + 1], sgt[g * 4 + 2], sgt[g * 4 + 3]);
        if (ov > best) { best = ov; arg = g; }   // first maximum
      }
      assign[(long)b * N + c] = arg;
      fg = best >= fg_thresh;
      bg = best < bg_hi && best >= bg_lo;
    }
    const unsigned long long mf = __ballot(fg), mb = __ballot(bg);
    if (lane == 0) { wave_fg[wave] = __popcll(mf); wave_bg[wave] = __popcll(mb); }
    __syncthreads();
    int off_f = base_fg, off_b = base_bg;
    for (int w = 0; w < wave; ++w) { off_f += wave_fg[w]; off_b += wave_bg[w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (fg) fg_list[(long)b * N + off_f + __popcll(mf & below)] = c;
    if (bg) bg_list[(long)b * N + off_b + __popcll(mb & below)] = c;
    __syncthreads();
    if (tid == 0) {
      int tf = 0, tb = 0;
      for (int w = 0; w < kPtThreads / 64; ++w) { tf += wave_fg[w]; tb += wave_bg[w]; }
      base_fg += tf; base_bg += tb;
    }
    __syncthreads();
  }
  if (tid == 0) { counts[b * 2] = base_fg; counts[b * 2 + 1] = base_bg; }
}

// bbox_transform.py:54-70 + the BBOX_NORMALIZE_* step (proposal_target_layer_cascade.py:111-116); log through double
// (correctly rounded binary32, the repo's declared semantics for exp / log in the box codec)
__device__ __forceinline__ void encode(const float* ex, const float* g, const float* mean, const float* stdv,
                                       int normalize, float* t) {
  const float ew = ex[2] - ex[0] + 1.f, eh = ex[3] - ex[1] + 1.f;
  const float ecx = ex[0] + 0.5f * ew, ecy = ex[1] + 0.5f * eh;
  const float gw = g[2] - g[0] + 1.f, gh = g[3] - g[1] + 1.f;
  const float gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
  t[0] = (gcx - ecx) / ew;
  t[1] = (gcy - ecy) / eh;
  t[2] = (float)log((double)(gw / ew));
  t[3] = (float)log((double)(gh / eh));
  if (normalize) {
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (t[k] - mean[k]) / stdv[k];
  }
}

struct PtParams {
  float mean[4], stdv[4], inside_w[4];
  int normalize, fg_per_image, n_out;
};

// grid = images, block = 256.  pos != NULL: host-chosen positions (first fg_n[b] index fg_list, the rest bg_list).
// pos == NULL: device mode from uniforms u_fg (B, N) / u_bg (B, n_out) (float64, [0, 1)).
// status[b]: 0 ok, 1 = neither foreground nor background candidates (the reference raises, :183-184).
__global__ __launch_bounds__(256) void pt_sample(const float* __restrict__ all_rois, const float* __restrict__ gt, int R,
                                                 int G, int gt_stride, const int* __restrict__ assign,
                                                 const int* __restrict__ fg_list, const int* __restrict__ bg_list,
                                                 const int* __restrict__ counts, const int* __restrict__ pos,
                                                 const int* __restrict__ fg_n_host, const double* __restrict__ u_fg,
                                                 const double* __restrict__ u_bg, PtParams P, float* __restrict__ rois_out,
                                                 float* __restrict__ labels_out, float* __restrict__ targets_out,
                                                 float* __restrict__ inside_out, float* __restrict__ outside_out,
                                                 int* __restrict__ status, int stage_keys) {
  extern __shared__ int sel[];   // n_out candidate indices (+ R + G staged keys when stage_keys)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = R + G, n = P.n_out;
  const int fg_cnt = counts[b * 2], bg_cnt = counts[b * 2 + 1];
  const int* fl = fg_list + (long)b * N;
  const int* bl = bg_list + (long)b * N;
  int fg_n;
  if (pos) {
    fg_n = fg_n_host[b];
    for (int j = tid; j < n; j += blockDim.x) sel[j] = j < fg_n ? fl[pos[(long)b * n + j]] : bl[pos[(long)b * n + j]];
  } else {
    const double* uf = u_fg + (long)b * N;
    const double* ub = u_bg + (long)b * n;
    if (fg_cnt > 0 && bg_cnt > 0) {
      fg_n = min(P.fg_per_image, fg_cnt);
      // the fg_n smallest keys of the foreground list (ties by position): a uniform subset without replacement
      // (the fg_cnt keys are staged in LDS once -- ADVICE r2: every thread used to re-read them from global memory in its
      // O(fg_cnt) ranking loop; the launch reserves N doubles behind the n_out selection slots)
      const double* ufs = uf;
      if (stage_keys) {
        double* st = reinterpret_cast<double*>(sel + ((n + 1) & ~1));
        for (int k = tid; k < fg_cnt; k += blockDim.x) st[k] = uf[k];
        __syncthreads();
        ufs = st;
      }
      for (int k = tid; k < fg_cnt; k += blockDim.x) {
        const double key = ufs[k];
        int rank = 0;
        for (int m = 0; m < fg_cnt; ++m) rank += (ufs[m] < key) || (ufs[m] == key && m < k);
        if (rank < fg_n) sel[rank] = fl[k];
      }
      for (int j = fg_n + tid; j < n; j += blockDim.x) sel[j] = bl[min((int)floor(ub[j - fg_n] * (double)bg_cnt), bg_cnt - 1)];
    } else if (fg_cnt > 0) {
      fg_n = n;
      for (int j = tid; j < n; j += blockDim.x) sel[j] = fl[min((int)floor(ub[j] * (double)fg_cnt), fg_cnt - 1)];
    } else if (bg_cnt > 0) {
      fg_n = 0;
      for (int j = tid; j < n; j += blockDim.x) sel[j] = bl[min((int)floor(ub[j] * (double)bg_cnt), bg_cnt - 1)];
    } else {
      fg_n = 0;
      for (int j = tid; j < n; j += blockDim.x) sel[j] = 0;
    }
  }
  if (tid == 0) status[b] = (fg_cnt == 0 && bg_cnt == 0) ? 1 : 0;
  __syncthreads();
  for (int j = tid; j < n; j += blockDim.x) {
    const int c = sel[j];
    float ex[4];
    if (c < R) {
      const float* r = all_rois + ((long)b * R + c) * 5;
      ex[0] = r[1]; ex[1] = r[2]; ex[2] = r[3]; ex[3] = r[4];
    } else {
      const float* g = gt + ((long)b * G + (c - R)) * gt_stride;
      ex[0] = g[0]; ex[1] = g[1]; ex[2] = g[2]; ex[3] = g[3];
    }
    const float* gsel = gt + ((long)b * G + assign[(long)b * N + c]) * gt_stride;
    const float label = j < fg_n ? gsel[4] : 0.f;   // background slots are clamped to 0 (:193-194)
    float t[4];
    encode(ex, gsel, P.mean, P.stdv, P.normalize, t);
    const long o = (long)b * n + j;
    rois_out[o * 5] = (float)b;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rois_out[o * 5 + 1 + k] = ex[k];
      const bool on = label > 0.f;
      const float iw = on ? P.inside_w[k] : 0.f;
      targets_out[o * 4 + k] = on ? t[k] : 0.f;
      inside_out[o * 4 + k] = iw;
      outside_out[o * 4 + k] = iw > 0.f ? 1.f : 0.f;
    }
    labels_out[o] = label;
  }
}

// grid = images, one wave.  gt (2, B, G, 6) [x1,y1,x2,y2,cls,track_id], nb (2, B) valid rows per frame.
__global__ __launch_bounds__(64) void tracking_target(const float* __restrict__ gt, const long* __restrict__ nb, int B, int G,
                                                      PtParams P, float* __restrict__ rois_out,
                                                      float* __restrict__ labels_out, float* __restrict__ targets_out,
                                                      float* __restrict__ inside_out, float* __restrict__ outside_out) {
  __shared__ float s0[kMaxGt * 6], s1[kMaxGt * 6];
  __shared__ int has0[kMaxGt], has1[kMaxGt], order0[kMaxGt], order1[kMaxGt];
  const int b = blockIdx.x, g = threadIdx.x;
  const float* f0 = gt + (long)b * G * 6;
  const float* f1 = gt + ((long)B + b) * G * 6;
  const int n0 = (int)nb[b], n1 = (int)nb[B + b];
  for (int i = g; i < G * 6; i += 64) { s0[i] = f0[i]; s1[i] = f1[i]; }
  __syncthreads();
  int h0 = 0, h1 = 0;
  if (g < G) {
    for (int m = 0; m < G; ++m) {
      h0 |= (g < n0 && m < n1 && s0[g * 6 + 5] == s1[m * 6 + 5]);
      h1 |= (g < n1 && m < n0 && s1[g * 6 + 5] == s0[m * 6 + 5]);
    }
    has0[g] = h0; has1[g] = h1;
  }
  const int c0 = __popcll(__ballot(h0)), c1 = __popcll(__ballot(h1));
  const bool ok = c0 > 0 && c1 > 0;
  __syncthreads();
  if (g < G) {
    // stable ascending rank of (matched ? track id : +inf) -- matched tracks first, sorted by id
    int r0 = 0, r1 = 0;
    const float big = 3.402823466e38f;
    const float k0 = has0[g] ? s0[g * 6 + 5] : big, k1 = has1[g] ? s1[g * 6 + 5] : big;
    for (int m = 0; m < G; ++m) {
      const float q0 = has0[m] ? s0[m * 6 + 5] : big, q1 = has1[m] ? s1[m * 6 + 5] : big;
      r0 += (q0 < k0) || (q0 == k0 && m < g);
      r1 += (q1 < k1) || (q1 == k1 && m < g);
    }
    order0[r0] = g; order1[r1] = g;
  }
  __syncthreads();
  if (g < G) {
    const long o = (long)b * G + g;
    const bool live0 = ok && g < c0, live1 = ok && g < c1;
    float a[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = live0 ? s0[order0[g] * 6 + k] : 0.f;
      c[k] = live1 ? s1[order1[g] * 6 + k] : 0.f;
    }
    const float label = live0 ? s0[order0[g] * 6 + 4] : 0.f;
    float t[4];
    encode(a, c, P.mean, P.stdv, P.normalize, t);
    rois_out[o * 5] = ok ? (float)b : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rois_out[o * 5 + 1 + k] = ok ? s0[g * 6 + k] : 0.f;   // RoIs stay in the original order (:171-185)
      const bool on = label > 0.f;
      const float iw = on ? P.inside_w[k] : 0.f;
      targets_out[o * 4 + k] = on ? t[k] : 0.f;
      inside_out[o * 4 + k] = iw;
      outside_out[o * 4 + k] = iw > 0.f ? 1.f : 0.f;
    }
    labels_out[o] = label;
  }
}

PtParams make_params(const float* mean4, const float* std4, const float* inside4, int normalize, int fg_per_image, int n_out) {
  PtParams P;
  for (int k = 0; k < 4; ++k) { P.mean[k] = mean4[k]; P.stdv[k] = std4[k]; P.inside_w[k] = inside4[k]; }
  P.normalize = normalize; P.fg_per_image = fg_per_image; P.n_out = n_out;
  return P;
}

}  // namespace

// Candidates = R proposals + G ground-truth boxes per image.  assign / fg_list / bg_list: int32 (images, R + G);
// counts: int32 (images, 2) = foreground / background candidates.
extern "C" int dtt_proposal_target_assign(const float* all_rois, const float* gt_boxes, int images, int num_rois,
                                          int num_gt, int gt_stride, float fg_thresh, float bg_thresh_hi,
                                          float bg_thresh_lo, int* assign, int* fg_list, int* bg_list, int* counts,
                                          void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(all_rois && gt_boxes && assign && fg_list && bg_list && counts, "proposal_target_assign: null pointer");
  DTT_REQUIRE(images > 0 && num_rois >= 0 && num_gt > 0 && num_gt <= kMaxGt && gt_stride >= 5,
              "proposal_target_assign: bad shape (at most %d ground-truth rows per image)", kMaxGt);
  hipLaunchKernelGGL(pt_assign, dim3(images), dim3(kPtThreads), 0, stream, all_rois, gt_boxes, num_rois, num_gt, gt_stride,
                     fg_thresh, bg_thresh_hi, bg_thresh_lo, assign, fg_list, bg_list, counts);
  DTT_CHECK_LAUNCH("pt_assign");
  return 1;
}

// pos_host_chosen / fg_n: int32 (images, n_out) / (images) on the DEVICE, or both NULL together with u_fg (images, R+G) and
// u_bg (images, n_out) float64 uniforms for the device-side choice.  Outputs: rois (images, n_out, 5), labels (images,
// n_out), targets / inside / outside (images, n_out, 4), status int32 (images).
extern "C" int dtt_proposal_target_sample(const float* all_rois, const float* gt_boxes, int images, int num_rois, int num_gt,
                                          int gt_stride, const int* assign, const int* fg_list, const int* bg_list,
                                          const int* counts, const int* pos, const int* fg_n, const double* u_fg,
                                          const double* u_bg, int n_out, int fg_per_image, const float* mean4_host,
                                          const float* std4_host, const float* inside4_host, int normalize, float* rois_out,
                                          float* labels_out, float* targets_out, float* inside_out, float* outside_out,
                                          int* status, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(all_rois && gt_boxes && assign && fg_list && bg_list && counts && rois_out && labels_out && targets_out &&
                  inside_out && outside_out && status && mean4_host && std4_host && inside4_host,
              "proposal_target_sample: null pointer");
  DTT_REQUIRE((pos && fg_n) || (!pos && !fg_n && u_fg && u_bg), "proposal_target_sample: pass positions + fg_n, or the two uniform arrays");
  DTT_REQUIRE(images > 0 && n_out > 0 && n_out <= 8192 && num_gt > 0 && num_gt <= kMaxGt, "proposal_target_sample: bad shape");
  const PtParams P = make_params(mean4_host, std4_host, inside4_host, normalize, fg_per_image, n_out);
  const size_t sel_bytes = (size_t)((n_out + 1) & ~1) * sizeof(int);
  const size_t key_bytes = (size_t)(num_rois + num_gt) * sizeof(double);
  const int stage_keys = !pos && sel_bytes + key_bytes <= 60 * 1024;
  hipLaunchKernelGGL(pt_sample, dim3(images), dim3(256), sel_bytes + (stage_keys ? key_bytes : 0), stream, all_rois, gt_boxes, num_rois, num_gt,
                     gt_stride, assign, fg_list, bg_list, counts, pos, fg_n, u_fg, u_bg, P, rois_out, labels_out, targets_out,
                     inside_out, outside_out, status, stage_keys);
  DTT_CHECK_LAUNCH("pt_sample");
  return 1;
}

// gt_boxes (2, images, num_gt, 6), num_boxes int64 (2, images).  Outputs as above with n_out = num_gt.
extern "C" int dtt_tracking_target(const float* gt_boxes, const long* num_boxes, int images, int num_gt,
                                   const float* mean4_host, const float* std4_host, const float* inside4_host,
                                   int normalize, float* rois_out, float* labels_out, float* targets_out, float* inside_out,
                                   float* outside_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gt_boxes && num_boxes && rois_out && labels_out && targets_out && inside_out && outside_out && mean4_host &&
                  std4_host && inside4_host, "tracking_target: null pointer");
  DTT_REQUIRE(images > 0 && num_gt > 0 && num_gt <= kMaxGt, "tracking_target: at most %d ground-truth rows per image", kMaxGt);
  const PtParams P = make_params(mean4_host, std4_host, inside4_host, normalize, 0, num_gt);
  hipLaunchKernelGGL(tracking_target, dim3(images), dim3(64), 0, stream, gt_boxes, num_boxes, images, num_gt, P, rois_out,
                     labels_out, targets_out, inside_out, outside_out);
  DTT_CHECK_LAUNCH("tracking_target");
  return 1;
}

// RPN anchor-target layer for gfx950.
//
// Replaces _AnchorTargetLayer.forward (reference rpn/anchor_target_layer.py:48-191): numpy meshgrid + H2D,
// a dense (B, N_inside, G) IoU tensor with ~10 temporaries (bbox_transform.py:208-254), Python per-image
// loops.  Here each anchor is one thread that recomputes its <= G IoUs in registers (nothing dense is
// materialised):
//   at_gt_max   : per-GT maximum IoU over the inside anchors (atomicMax on the float bits)
//   at_assign   : labels {1, 0, -1}, arg-max GT, per-image fg / bg counts
//   (host draws numpy permutations exactly like the reference, anchor_target_layer.py:124-141)
//   at_disable  : applies the host's disable lists
//   at_finish   : regression targets (bbox_transform.py:36-75), weights and the four output layouts
//                 (anchor_target_layer.py:168-189)
// IoU / encode arithmetic is binary32 in the reference's operation order, FP contraction off; log is
// correctly rounded via double (declared semantics, see oracle/rpn_oracle.py).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct AtGeom {
  int B, G, A, H, W, K, n, stride;
  int im_h, im_w;  // long(im_info[0][0]), long(im_info[0][1])  (image 0 only, anchor_target_layer.py:85-86)
  const float* im_info;   // device mode: the kernels read row 0 themselves (no host copy of im_info); else null
};

// (image height, width) the "inside the image" test uses: truncated like the reference's long(im_info[0][.])
__device__ __forceinline__ void image_hw(const AtGeom& g, int& h, int& w) {
  if (g.im_info) { h = (int)(long)g.im_info[0]; w = (int)(long)g.im_info[1]; }
  else { h = g.im_h; w = g.im_w; }
}

__device__ __forceinline__ void anchor_of(const float* __restrict__ base, const AtGeom& g, int t, float& x1, float& y1,
                                          float& x2, float& y2) {
  const int k = t / g.A, a = t - k * g.A;
  const int h = k / g.W, w = k - h * g.W;
  const float sx = (float)(w * g.stride), sy = (float)(h * g.stride);
  x1 = base[a * 4 + 0] + sx; y1 = base[a * 4 + 1] + sy;
  x2 = base[a * 4 + 2] + sx; y2 = base[a * 4 + 3] + sy;
}

__device__ __forceinline__ bool inside(const AtGeom& g, float x1, float y1, float x2, float y2) {
  int im_h, im_w;
  image_hw(g, im_h, im_w);
  return x1 >= 0.f && y1 >= 0.f && x2 < (float)im_w && y2 < (float)im_h;
}

// bbox_transform.py:228-254 for one (anchor, gt) pair
__device__ __forceinline__ float overlap(float ax1, float ay1, float ax2, float ay2, const float* __restrict__ gt) {
  const float gx = gt[2] - gt[0] + 1.f, gy = gt[3] - gt[1] + 1.f;
  const float g_area = gx * gy;
  const float ax = ax2 - ax1 + 1.f, ay = ay2 - ay1 + 1.f;
  const float a_area = ax * ay;
  float iw = fminf(ax2, gt[2]) - fmaxf(ax1, gt[0]) + 1.f;
  if (iw < 0.f) iw = 0.f;
  float ih = fminf(ay2, gt[3]) - fmaxf(ay1, gt[1]) + 1.f;
  if (ih < 0.f) ih = 0.f;
  const float ua = a_area + g_area - (iw * ih);
  float ov = iw * ih / ua;
  if (gx == 1.f && gy == 1.f) ov = 0.f;
  if (ax == 1.f && ay == 1.f) ov = -1.f;
  return ov;
}

// gt_max bits initialised to -1.0f by the launcher (memset pattern via kernel below)
__global__ void at_init(int* __restrict__ gt_max_bits, int n, int* __restrict__ counts, int ncounts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gt_max_bits[i] = (int)0xBF800000;  // -1.0f: below every IoU >= 0 as a signed int
  if (i < ncounts) counts[i] = 0;
}

// Per-GT maximum over the inside anchors.  IoUs >= 0 order like their bit patterns, so the maximum is an integer
// max: reduced across the wave by shuffles, across the workgroup through LDS, and only then one atomic per
// (workgroup, gt) reaches memory -- the per-anchor atomics of a direct translation all hit the same B*G words.
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(kThreads) void at_gt_max(const float* __restrict__ gt_boxes, const float* __restrict__ base,
                                                      AtGeom g, int* __restrict__ gt_max_bits) {
  constexpr int kNone = (int)0xBF800000;  // -1.0f
  __shared__ int wmax[kThreads / 64][32];
  const int t = blockIdx.x * kThreads + threadIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
  bool live = t < g.n;
  if (live) {
    anchor_of(base, g, t, x1, y1, x2, y2);
    live = inside(g, x1, y1, x2, y2);
  }
  for (int j0 = 0; j0 < g.G; j0 += 32) {
    const int nj = min(32, g.G - j0);
    for (int j = 0; j < nj; ++j) {
      int v = kNone;
      if (live) {
        const float ov = overlap(x1, y1, x2, y2, gt_boxes + ((long)b * g.G + j0 + j) * 5);
        if (ov >= 0.f) v = __float_as_int(ov + 0.f);
      }
      v = wave_max_i32(v);
      if (lane == 0) wmax[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < nj) {
      int v = wmax[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < kThreads / 64; ++w) v = max(v, wmax[w][threadIdx.x]);
      if (v != kNone) atomicMax(&gt_max_bits[b * g.G + j0 + threadIdx.x], v);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void at_assign(const float* __restrict__ gt_boxes, const float* __restrict__ base,
                                                      AtGeom g, const int* __restrict__ gt_max_bits, float neg_thr,
                                                      float pos_thr, int clobber, int* __restrict__ labels,
                                                      int* __restrict__ argmax_gt, int* __restrict__ counts) {
  const int t = blockIdx.x * kThreads + threadIdx.x, b = blockIdx.y;
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
  int label = -1, amax = 0;
  bool live = t < g.n;
  if (live) {
    anchor_of(base, g, t, x1, y1, x2, y2);
    if (inside(g, x1, y1, x2, y2)) {
      float best = -INFINITY;
      bool is_gt_best = false;
      for (int j = 0; j < g.G; ++j) {
        const float ov = overlap(x1, y1, x2, y2, gt_boxes + ((long)b * g.G + j) * 5);
        if (ov > best) { best = ov; amax = j; }  // first maximum, as torch.max(dim)
        float gm = __int_as_float(gt_max_bits[b * g.G + j]);
        if (gm == 0.f) gm = 1e-5f;                // anchor_target_layer.py:104
        if (ov == gm) is_gt_best = true;          // anchor_target_layer.py:105
      }
      if (!clobber && best < neg_thr) label = 0;
      if (is_gt_best) label = 1;
      if (best >= pos_thr) label = 1;
      if (clobber && best < neg_thr) label = 0;
    }
    labels[(long)b * g.n + t] = label;
    argmax_gt[(long)b * g.n + t] = amax;
  }
  // fg / bg counts: one atomic per wave, not per anchor
  const unsigned long long fg = __ballot(live && label == 1), bg = __ballot(live && label == 0);
  if ((threadIdx.x & 63) == 0) {
    if (fg) atomicAdd(&counts[b * 2 + 0], __popcll(fg));
    if (bg) atomicAdd(&counts[b * 2 + 1], __popcll(bg));
  }
}

__global__ void at_disable(int* __restrict__ labels, const int* __restrict__ disable,
                           const int* __restrict__ offsets, int total_anchors) {
  const int b = blockIdx.y;
  const int beg = offsets[b], end = offsets[b + 1];
  for (int i = beg + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
    labels[(long)b * total_anchors + disable[i]] = -1;
}

__global__ __launch_bounds__(kThreads) void at_finish(const float* __restrict__ gt_boxes, const float* __restrict__ base,
                                                      AtGeom g, const int* __restrict__ labels,
                                                      const int* __restrict__ argmax_gt, float inside_w, float pos_w,
                                                      float neg_w, const float* __restrict__ dev_weights,
                                                      float* __restrict__ labels_out,
                                                      float* __restrict__ targets, float* __restrict__ in_w,
                                                      float* __restrict__ out_w) {
  const int t = blockIdx.x * kThreads + threadIdx.x, b = blockIdx.y;
  if (t >= g.n) return;
  if (dev_weights) { pos_w = dev_weights[0]; neg_w = dev_weights[1]; }   // device mode: written by at_subsample
  const int k = t / g.A, a = t - k * g.A;
  float x1, y1, x2, y2;
  anchor_of(base, g, t, x1, y1, x2, y2);
  const int label = labels[(long)b * g.n + t];
  float tx = 0.f, ty = 0.f, tw = 0.f, th = 0.f;
  if (inside(g, x1, y1, x2, y2)) {
    const float* gt = gt_boxes + ((long)b * g.G + argmax_gt[(long)b * g.n + t]) * 5;
    const float ew = x2 - x1 + 1.0f, eh = y2 - y1 + 1.0f;
    const float ecx = x1 + 0.5f * ew, ecy = y1 + 0.5f * eh;
    const float gw = gt[2] - gt[0] + 1.0f, gh = gt[3] - gt[1] + 1.0f;
    const float gcx = gt[0] + 0.5f * gw, gcy = gt[1] + 0.5f * gh;
    tx = (gcx - ecx) / ew;
    ty = (gcy - ecy) / eh;
    tw = (float)log((double)(gw / ew));
    th = (float)log((double)(gh / eh));
  }
  // labels: (B, K*A) -> view (B,H,W,A) -> permute (0,3,1,2) -> (B,1,A*H,W)   (anchor_target_layer.py:171-173)
  labels_out[((long)b * g.A + a) * g.K + k] = (float)label;
  const float iw = label == 1 ? inside_w : 0.f;
  const float ow = label == 1 ? pos_w : (label == 0 ? neg_w : 0.f);
  const long o = ((long)b * 4 * g.A + 4 * a) * g.K + k;
  targets[o] = tx; targets[o + g.K] = ty; targets[o + 2L * g.K] = tw; targets[o + 3L * g.K] = th;
  in_w[o] = iw; in_w[o + g.K] = iw; in_w[o + 2L * g.K] = iw; in_w[o + 3L * g.K] = iw;
  out_w[o] = ow; out_w[o + g.K] = ow; out_w[o + 2L * g.K] = ow; out_w[o + 3L * g.K] = ow;
}

// Device-mode subsampling (anchor_target_layer.py:118-141 without the host): one workgroup per image.  The reference disables a
// uniformly random subset of the foreground anchors beyond num_fg and of the background anchors beyond
// rpn_batchsize - sum_fg (numpy permutations of the index lists).  Here every anchor carries a 32-bit random key the caller drew
// WITHOUT looking at the labels, and the k candidates of a class with the smallest (key, index) stay: a uniform subset of
// exactly k, the same distribution, nothing read back.  The k-th smallest key is found by a radix select (four 8-bit passes over
// the image's labels and keys, a 256-bin histogram in LDS per class: foreground and background are selected in the same passes);
// candidates with exactly that key -- rare -- are kept in index order.
// after[b] = (fg, bg) counts left; the workgroup of the LAST image also writes the two outside weights the reference derives from
// that image's counts (anchor_target_layer.py:143-154): weights[0] positive, [1] negative.
constexpr int kSubThreads = 1024;
constexpr int kBatch = 16;   // label / key pairs in flight per thread
// The fast path (images of up to kCache * 1024 = 32 768 anchors: 30 552 at 600 x 1067): ONE pass over memory.  A thread keeps its
// kCache label / key pairs in registers; a 4096-bin histogram of the keys' top 12 bits per class finds the bin that holds the
// quota-th smallest key (a parallel scan over the bins, not a serial walk), the handful of candidates of that bin (n / 4096 on
// average) go to a list in LDS where each is ranked against the others by (key, index), and the labels beyond the threshold pair
// are cleared from the registers.  A bin with more than kList candidates (keys that are not random) falls back to the radix
// select below -- the same subset either way: the quota smallest (key, index) pairs.
constexpr int kCache = 32, kBins = 4096, kList = 1024;

struct SubShared {
  int hist[2][kBins];
  unsigned list_key[2][kList];
  int list_idx[2][kList];
  int wsum[2][8];
  int ctl[2][3];       // the bin, how many of its candidates stay, how many it holds
  int lcount[2];
  unsigned thr_key[2];
  int thr_idx[2];
};

__device__ bool at_subsample_fast(SubShared& S, int* __restrict__ lab, const unsigned* __restrict__ key, int n, bool sel0, bool sel1,
                                  bool cut0, bool cut1, int quota0, int quota1) {
  const int tid = threadIdx.x;
  auto sel = [&](int c) { return c ? sel1 : sel0; };
  auto cut = [&](int c) { return c ? cut1 : cut0; };
  int cs[kCache];
  unsigned ks[kCache];
#pragma unroll
  for (int u = 0; u < kCache; ++u) {
    const int i = tid + u * kSubThreads;
    cs[u] = i < n ? lab[i] : -1;
    ks[u] = i < n ? key[i] : 0u;
  }
  for (int i = tid; i < 2 * kBins; i += kSubThreads) (&S.hist[0][0])[i] = 0;
  if (tid < 2) S.lcount[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kCache; ++u)
    if (cs[u] >= 0 && sel(cs[u])) atomicAdd(&S.hist[cs[u]][ks[u] >> 20], 1);
  __syncthreads();
  {
    // class c = tid / 512 scans its 4096 bins, 8 per thread
    const int c = tid >> 9, t = tid & 511, lane = tid & 63, wv = t >> 6;
    int own = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) own += S.hist[c][t * 8 + j];
    int incl = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) S.wsum[c][wv] = incl;
    __syncthreads();
    int before = incl - own;
    for (int w = 0; w < wv; ++w) before += S.wsum[c][w];
    const int rem = c ? quota1 : quota0;
    if (sel(c) && before < rem && before + own >= rem) {
      int cum = before, d = t * 8;
      for (; d < t * 8 + 7; ++d) {
        if (cum + S.hist[c][d] >= rem) break;
        cum += S.hist[c][d];
      }
      S.ctl[c][0] = d; S.ctl[c][1] = rem - cum; S.ctl[c][2] = S.hist[c][d];
    }
  }
  __syncthreads();
  if ((sel0 && S.ctl[0][2] > kList) || (sel1 && S.ctl[1][2] > kList)) return false;   // (uniform: read from LDS)
#pragma unroll
  for (int u = 0; u < kCache; ++u) {
    const int c = cs[u];
    if (c >= 0 && sel(c) && (int)(ks[u] >> 20) == S.ctl[c][0]) {
      const int pos = atomicAdd(&S.lcount[c], 1);
      S.list_key[c][pos] = ks[u]; S.list_idx[c][pos] = tid + u * kSubThreads;
    }
  }
  __syncthreads();
  {
    const int c = tid >> 9;
    if (sel(c)) {
      const int m = S.ctl[c][2], want = S.ctl[c][1] - 1;
      for (int e = tid & 511; e < m; e += 512) {
        const unsigned k = S.list_key[c][e];
        const int i = S.list_idx[c][e];
        int rank = 0;
        for (int o = 0; o < m; ++o) {
          const unsigned ko = S.list_key[c][o];
          rank += (ko < k || (ko == k && S.list_idx[c][o] < i)) ? 1 : 0;
        }
        if (rank == want) { S.thr_key[c] = k; S.thr_idx[c] = i; }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kCache; ++u) {
    const int i = tid + u * kSubThreads, c = cs[u];
    if (i >= n || c < 0 || !cut(c)) continue;
    const bool keep = sel(c) && (ks[u] < S.thr_key[c] || (ks[u] == S.thr_key[c] && i <= S.thr_idx[c]));
    if (!keep) lab[i] = -1;
  }
  return true;
}

__global__ __launch_bounds__(kSubThreads) void at_subsample(int* __restrict__ labels, const unsigned* __restrict__ keys, int n,
                                                            int batch, const int* __restrict__ counts, int rpn_batchsize,
                                                            int num_fg, float positive_weight, int* __restrict__ after,
                                                            float* __restrict__ weights, int no_fast) {
  // per class c (0 background, 1 foreground): hist[c][256]; ctl[c] = {digit, remaining, equal}
  __shared__ SubShared S;
  int (*hist)[256] = reinterpret_cast<int (*)[256]>(&S.hist[0][0]);      // (the radix select's 2 x 256 bins alias the fast path's histogram)
  int (*ctl)[3] = S.ctl;
  int* wave_cnt = &S.list_idx[0][0];
  const int b = blockIdx.x, tid = threadIdx.x;
  int* lab = labels + (long)b * n;
  const unsigned* key = keys + (long)b * n;
  const int sum_fg = counts[b * 2], sum_bg = counts[b * 2 + 1];
  const int num_bg = rpn_batchsize - sum_fg;   // the fg count BEFORE subsampling, as in the reference (:133)
  // class c is cut down to quota[c] when it has more candidates than that (quota <= 0: every candidate goes)
  const int quota[2] = {num_bg, num_fg};
  const bool cut[2] = {sum_bg > num_bg, sum_fg > num_fg};
  const bool sel[2] = {cut[0] && quota[0] > 0, cut[1] && quota[1] > 0};   // a k-th smallest key has to be found
  unsigned prefix[2] = {0u, 0u};
  int rem[2] = {quota[0], quota[1]}, equal[2] = {0, 0};
  unsigned mask = 0;
  static_assert(kSubThreads == 1024 && kBins == 8 * 512, "at_subsample_fast: two classes x 512 threads x 8 bins");
  const bool fast = (cut[0] || cut[1]) && n <= kCache * kSubThreads && !no_fast && at_subsample_fast(S, lab, key, n, sel[0], sel[1], cut[0], cut[1], quota[0], quota[1]);
  if (fast) {
    // (labels cleared; the counts below are all that is left)
  } else {
  if (sel[0] || sel[1]) {
    // radix select, both classes in the same four passes over the image's labels / keys
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 512) hist[tid >> 8][tid & 255] = 0;
      __syncthreads();
      // (one CU walks the image: kBatch label / key pairs per thread are requested before any is used -- the pass is a chain of
      //  memory round trips otherwise, 30 of them at 30 552 anchors)
      for (int i0 = tid; i0 < n; i0 += kBatch * kSubThreads) {
        int cs[kBatch];
        unsigned ks[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int i = min(i0 + u * kSubThreads, n - 1);
          cs[u] = lab[i]; ks[u] = key[i];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
          const int c = cs[u];
          if (i0 + u * kSubThreads < n && c >= 0 && (c ? sel[1] : sel[0]) && (ks[u] & mask) == (c ? prefix[1] : prefix[0]))
            atomicAdd(&hist[c][(ks[u] >> shift) & 255u], 1);
        }
      }
      __syncthreads();
      if ((tid == 0 || tid == 64) && sel[tid >> 6]) {   // one lane per class walks its 256 bins
        const int c = tid >> 6;
        int cum = 0, d = 0;
        for (; d < 255; ++d) {
          if (cum + hist[c][d] >= rem[c]) break;
          cum += hist[c][d];
        }
        ctl[c][0] = d; ctl[c][1] = rem[c] - cum; ctl[c][2] = hist[c][d];
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (sel[c]) { prefix[c] |= (unsigned)ctl[c][0] << shift; rem[c] = ctl[c][1]; equal[c] = ctl[c][2]; }
      mask |= 255u << shift;
      __syncthreads();
    }
  }
  // prefix[c] = the k-th smallest key T of class c; rem[c] of the equal[c] candidates with key == T stay (the first in index order)
  const bool ordered[2] = {sel[0] && rem[0] < equal[0], sel[1] && rem[1] < equal[1]};
  if (cut[0] || cut[1]) {
    for (int i0 = tid; i0 < n; i0 += kBatch * kSubThreads) {
      int cs[kBatch];
      unsigned ks[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = min(i0 + u * kSubThreads, n - 1);
        cs[u] = lab[i]; ks[u] = key[i];
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = i0 + u * kSubThreads, c = cs[u];
        if (i >= n || c < 0 || !(c ? cut[1] : cut[0])) continue;
        if (!(c ? sel[1] : sel[0]) || ks[u] > (c ? prefix[1] : prefix[0])) lab[i] = -1;
      }
    }
  }
  for (int c = 0; c < 2; ++c) {
    if (!ordered[c]) continue;   // (uniform)
    int running = 0;             // equal-key candidates seen in earlier chunks
    for (int i0 = 0; i0 < n; i0 += kSubThreads) {
      const int i = i0 + tid;
      const bool eq = i < n && lab[i] == c && key[i] == prefix[c];
      const unsigned long long m = __ballot(eq);
      __syncthreads();
      if ((tid & 63) == 0) wave_cnt[tid >> 6] = __popcll(m);
      __syncthreads();
      int before = running, total = 0;
      for (int w = 0; w < kSubThreads / 64; ++w) {
        if (w < (tid >> 6)) before += wave_cnt[w];
        total += wave_cnt[w];
      }
      const int rank = before + __popcll(m & ((1ull << (tid & 63)) - 1ull));
      if (eq && rank >= rem[c]) lab[i] = -1;
      running += total;
    }
  }
  }
  if (tid == 0) {
    const int fg_left = cut[1] ? max(num_fg, 0) : sum_fg, bg_left = cut[0] ? max(num_bg, 0) : sum_bg;
    after[b * 2] = fg_left; after[b * 2 + 1] = bg_left;
    if (b == batch - 1) {
      const int num_examples = fg_left + bg_left;
      float w_pos = num_examples > 0 ? 1.0f / (float)num_examples : INFINITY, w_neg = w_pos;
      if (positive_weight >= 0.f) {
        w_pos = fg_left > 0 ? positive_weight / (float)fg_left : INFINITY;
        w_neg = bg_left > 0 ? (1.0f - positive_weight) / (float)bg_left : INFINITY;
      }
      weights[0] = w_pos; weights[1] = w_neg;
    }
  }
}

}  // namespace

// Host-side geometry needs im_info[0]; it is passed by value to keep the call free of D2H copies.
extern "C" int dtt_anchor_target_assign(const float* gt_boxes, int im_h0, int im_w0, const float* anchors,
                                           int batch, int num_gt, int num_anchors, int height, int width,
                                           int feat_stride, float negative_overlap, float positive_overlap,
                                           int clobber_positives, int* labels, int* argmax_gt, int* counts,
                                           int* gt_max_scratch, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gt_boxes && anchors && labels && argmax_gt && counts && gt_max_scratch, "anchor_target: null pointer");
  DTT_REQUIRE(batch > 0 && num_gt > 0 && num_anchors > 0 && height > 0 && width > 0, "anchor_target: bad shape");
  AtGeom g;
  g.B = batch; g.G = num_gt; g.A = num_anchors; g.H = height; g.W = width; g.K = height * width;
  g.n = g.K * g.A; g.stride = feat_stride; g.im_h = im_h0; g.im_w = im_w0; g.im_info = nullptr;
  const int ninit = batch * num_gt > batch * 2 ? batch * num_gt : batch * 2;
  hipLaunchKernelGGL(at_init, dim3(dtt_cdiv(ninit, 256)), dim3(256), 0, stream, gt_max_scratch, batch * num_gt, counts,
                     batch * 2);
  dim3 grid(dtt_cdiv(g.n, kThreads), batch);
  hipLaunchKernelGGL(at_gt_max, grid, dim3(kThreads), 0, stream, gt_boxes, anchors, g, gt_max_scratch);
  hipLaunchKernelGGL(at_assign, grid, dim3(kThreads), 0, stream, gt_boxes, anchors, g, gt_max_scratch,
                     negative_overlap, positive_overlap, clobber_positives, labels, argmax_gt, counts);
  DTT_CHECK_LAUNCH("anchor_target assign");
  return 1;
}

extern "C" int dtt_anchor_target_disable(int* labels, const int* disable, const int* disable_offsets, int batch,
                                         int total_anchors, void* stream_) {
  DTT_REQUIRE(labels && disable && disable_offsets && batch > 0, "anchor_target disable: null pointer");
  hipLaunchKernelGGL(at_disable, dim3(8, batch), dim3(256), 0, static_cast<hipStream_t>(stream_), labels, disable,
                     disable_offsets, total_anchors);
  DTT_CHECK_LAUNCH("anchor_target disable");
  return 1;
}

extern "C" int dtt_anchor_target_finish(const float* gt_boxes, int im_h0, int im_w0, const float* anchors,
                                           const int* labels_in, const int* argmax_gt, int batch, int num_gt,
                                           int num_anchors, int height, int width, int feat_stride,
                                           float inside_weight, float positive_weight, float negative_weight,
                                           float* labels_out, float* bbox_targets, float* bbox_inside_weights,
                                           float* bbox_outside_weights, void* stream_) {
  DTT_REQUIRE(gt_boxes && anchors && labels_in && argmax_gt && labels_out && bbox_targets && bbox_inside_weights &&
                  bbox_outside_weights,
              "anchor_target finish: null pointer");
  AtGeom g;
  g.B = batch; g.G = num_gt; g.A = num_anchors; g.H = height; g.W = width; g.K = height * width;
  g.n = g.K * g.A; g.stride = feat_stride; g.im_h = im_h0; g.im_w = im_w0; g.im_info = nullptr;
  hipLaunchKernelGGL(at_finish, dim3(dtt_cdiv(g.n, kThreads), batch), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream_), gt_boxes, anchors, g, labels_in, argmax_gt, inside_weight,
                     positive_weight, negative_weight, nullptr, labels_out, bbox_targets, bbox_inside_weights,
                     bbox_outside_weights);
  DTT_CHECK_LAUNCH("anchor_target finish");
  return 1;
}

// The whole layer without a host read (cfg.TRAIN.SAMPLER_RNG = "device"): assign, subsample by the caller's random keys
// (at_subsample), finish -- im_info stays on the device (row 0 is read by the kernels), the per-image counts and the outside
// weights never leave it.  keys: (batch, K*A) uint32 drawn by the caller without knowledge of the labels.  positive_weight < 0:
// uniform weighting (anchor_target_layer.py:143-147).  Scratch: labels / argmax_gt (batch, K*A) int32, counts (batch, 4) int32
// ([0, 2 batch) before, [2 batch, 4 batch) after subsampling), gt_max_scratch (batch * num_gt) int32, weights (2) float.
extern "C" int dtt_anchor_target_device(const float* gt_boxes, const float* im_info, const float* anchors,
                                        const unsigned* keys, int batch, int num_gt, int num_anchors, int height, int width,
                                        int feat_stride, int rpn_batchsize, int num_fg, float negative_overlap,
                                        float positive_overlap, int clobber_positives, float inside_weight,
                                        float positive_weight, int* labels, int* argmax_gt, int* counts,
                                        int* gt_max_scratch, float* weights, float* labels_out, float* bbox_targets,
                                        float* bbox_inside_weights, float* bbox_outside_weights, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gt_boxes && im_info && anchors && keys && labels && argmax_gt && counts && gt_max_scratch && weights && labels_out &&
                  bbox_targets && bbox_inside_weights && bbox_outside_weights, "anchor_target (device): null pointer");
  DTT_REQUIRE(batch > 0 && num_gt > 0 && num_anchors > 0 && height > 0 && width > 0 && rpn_batchsize > 0 && num_fg >= 0,
              "anchor_target (device): bad shape");
  AtGeom g;
  g.B = batch; g.G = num_gt; g.A = num_anchors; g.H = height; g.W = width; g.K = height * width;
  g.n = g.K * g.A; g.stride = feat_stride; g.im_h = 0; g.im_w = 0; g.im_info = im_info;
  const int ninit = batch * num_gt > batch * 2 ? batch * num_gt : batch * 2;
  dtt_prof_begin("anchor_target_op", stream);   // (event tag: the whole layer in device mode, five launches)
  hipLaunchKernelGGL(at_init, dim3(dtt_cdiv(ninit, 256)), dim3(256), 0, stream, gt_max_scratch, batch * num_gt, counts, batch * 2);
  dim3 grid(dtt_cdiv(g.n, kThreads), batch);
  hipLaunchKernelGGL(at_gt_max, grid, dim3(kThreads), 0, stream, gt_boxes, anchors, g, gt_max_scratch);
  hipLaunchKernelGGL(at_assign, grid, dim3(kThreads), 0, stream, gt_boxes, anchors, g, gt_max_scratch, negative_overlap,
                     positive_overlap, clobber_positives, labels, argmax_gt, counts);
  const int no_fast = getenv("DTT_AT_SUBSAMPLE_SLOW") ? 1 : 0;             // developer / test switch: the radix select only
  hipLaunchKernelGGL(at_subsample, dim3(batch), dim3(kSubThreads), 0, stream, labels, keys, g.n, batch, counts, rpn_batchsize,
                     num_fg, positive_weight, counts + 2 * batch, weights, no_fast);
  hipLaunchKernelGGL(at_finish, grid, dim3(kThreads), 0, stream, gt_boxes, anchors, g, labels, argmax_gt, inside_weight, 0.f, 0.f,
                     weights, labels_out, bbox_targets, bbox_inside_weights, bbox_outside_weights);
  dtt_prof_end("anchor_target_op", stream);
  DTT_CHECK_LAUNCH("anchor_target (device)");
  return 1;
}

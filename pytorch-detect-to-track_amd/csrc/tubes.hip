// Zero-jump Viterbi tube linking for gfx950 (the "track" half of Detect-to-Track's post-processing).
//
// Replaces VideoPostProcessor._make_tubes / _zero_jump_link / _score_of_edge (reference
// lib/model/utils/tracking_utils.py:86-124, 127-264, 268-290): per class the reference runs, in Python, a per-frame
// NMS, then for each of up to 25 paths a Viterbi pass over all frames whose every step rebuilds the N1 x N2 edge
// scores (row by row in a Python loop), two IoU matrices against the frame's tracklets and a matrix product -- tens of
// tiny kernels and several host syncs per frame per path per class.  Here, for all classes at once:
//   1. tube_frame_nms_kernel  one workgroup per (class, frame): greedy NMS in LDS in the given priority order, the
//                             first max_per_image survivors are kept (tracking_utils.py:108-115);
//   2. tube_bonus_kernel      one workgroup per (class, frame pair): rounded IoUs of the kept boxes against the
//                             frame's tracklets, their product over tracklets -> one 32-bit "linked by a tracklet"
//                             mask per box (the + 1.0 term of _score_of_edge); computed ONCE, not once per path,
//                             because removing a path's boxes changes neither scores nor masks;
//   3. tube_viterbi_kernel    one wave per class: all K = min_t n_t paths back to back -- forward max-plus recursion
//                             over the alive boxes (lane = box, the previous frame's scores travel by shuffles),
//                             back-pointers in LDS, backtrace, removal of the path's boxes from the alive masks.
// Arithmetic order is the reference's: ((s1 + s2) [+ 1.0]) + D[t+1], maxima and the start box take the lowest index
// among equals (PyTorch's CPU semantics, pinned by tests/golden/tubes.npz).
#include "common.h"
#include "nms_small.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxK = 32;      // boxes kept per frame (reference: 25) -- one 32-bit mask word
constexpr int kMaxDets = 1024; // detections per frame before NMS
constexpr int kMaxTrk = 512;   // tracklets per frame

struct TubeDims {
  int P, F, T, Nmax, Mmax, K, det_stride;
};

// grid (T, P).  dets [P][F][Nmax][det_stride], n [P][F]  ->  kbox [P][T][kMaxK][4], kscore [P][T][kMaxK], kn [P][T]
__global__ __launch_bounds__(kThreads) void tube_frame_nms_kernel(const float* __restrict__ dets,
                                                                  const int* __restrict__ n_in, TubeDims d,
                                                                  float nms_thresh, int P2, float* __restrict__ kbox,
                                                                  float* __restrict__ kscore, int* __restrict__ kn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = P2 / 64;
  float4* box = reinterpret_cast<float4*>(smem);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(box + P2);
  unsigned short* kept = reinterpret_cast<unsigned short*>(mask + (size_t)P2 * W);
  int* ctl = reinterpret_cast<int*>(kept + P2);
  const int t = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
  const int n = min(n_in[p * d.F + t], d.Nmax);
  const float* src = dets + ((long)p * d.F + t) * d.Nmax * d.det_stride;
  const long o = (long)p * d.T + t;
  if (n <= 0) {
    if (tid == 0) kn[o] = 0;
    return;
  }
  for (int i = tid; i < n; i += kThreads) {
    const float* r = src + (long)i * d.det_stride;
    box[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (tid == 0) ctl[1] = 0;
  __syncthreads();
  dtt_small_nms::lds_mask_and_sweep(box, n, W, nms_thresh, mask, kept, ctl, tid, kThreads);
  const int nk = min(ctl[1], d.K);
  if (tid == 0) kn[o] = nk;
  if (tid < nk) {
    const int i = kept[tid];
    const float4 b = box[i];
    float* ob = kbox + (o * kMaxK + tid) * 4;
    ob[0] = b.x; ob[1] = b.y; ob[2] = b.z; ob[3] = b.w;
    kscore[o * kMaxK + tid] = src[(long)i * d.det_stride + 4];
  }
}

// bbox_overlaps of model/rpn/bbox_transform.py:175-206 for one pair, then torch.round (half to even)
__device__ __forceinline__ float rounded_overlap(const float4 a, const float4 g) {
  const float g_area = (g.z - g.x + 1.f) * (g.w - g.y + 1.f);
  const float a_area = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
  float iw = fminf(a.z, g.z) - fmaxf(a.x, g.x) + 1.f;
  if (iw < 0.f) iw = 0.f;
  float ih = fminf(a.w, g.w) - fmaxf(a.y, g.y) + 1.f;
  if (ih < 0.f) ih = 0.f;
  const float ua = a_area + g_area - iw * ih;
  return rintf(iw * ih / ua);
}

// grid (T - 1, P).  bonus [P][T][kMaxK]: bit j of word (t, i) = boxes (t, i) and (t + 1, j) are linked by a tracklet of
// frame t (sum over tracklets of round(IoU(box_t_i, trk0_m)) * round(IoU(box_t+1_j, trk1_m)) > 0), provided frames t
// and t + 1 both carry tracklets (tracking_utils.py:280).
__global__ __launch_bounds__(kThreads) void tube_bonus_kernel(const float* __restrict__ kbox, const int* __restrict__ kn,
                                                              const float* __restrict__ trk, const int* __restrict__ m_in,
                                                              TubeDims d, unsigned* __restrict__ bonus) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* r1 = reinterpret_cast<float*>(smem);          // [kMaxK][Mmax]
  float* r2 = r1 + (size_t)kMaxK * d.Mmax;             // [kMaxK][Mmax]
  const int t = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
  const long o = (long)p * d.T + t;
  const int m0 = trk ? m_in[p * d.F + t] : -1, m1 = trk ? m_in[p * d.F + t + 1] : -1;
  const int n1 = kn[o], n2 = kn[o + 1];
  if (m0 <= 0 || m1 < 0) {   // m1 == 0 cannot happen for a frame that "has" tracklets; < 0 means None
    if (tid < kMaxK) bonus[o * kMaxK + tid] = 0u;
    return;
  }
  const int M = min(m0, d.Mmax);
  const float4* t0 = reinterpret_cast<const float4*>(trk + (((long)p * d.F + t) * 2 + 0) * d.Mmax * 4);
  const float4* t1 = reinterpret_cast<const float4*>(trk + (((long)p * d.F + t) * 2 + 1) * d.Mmax * 4);
  const float4* b1 = reinterpret_cast<const float4*>(kbox + o * kMaxK * 4);
  const float4* b2 = reinterpret_cast<const float4*>(kbox + (o + 1) * kMaxK * 4);
  for (int idx = tid; idx < n1 * M; idx += kThreads) {
    const int i = idx / M, mm = idx - i * M;
    r1[i * d.Mmax + mm] = rounded_overlap(b1[i], t0[mm]);
  }
  for (int idx = tid; idx < n2 * M; idx += kThreads) {
    const int j = idx / M, mm = idx - j * M;
    r2[j * d.Mmax + mm] = rounded_overlap(b2[j], t1[mm]);
  }
  __syncthreads();
  // one (i, j) pair per thread slot: 32 x 32 pairs over 256 threads
  for (int i0 = 0; i0 < kMaxK; i0 += kThreads / kMaxK) {
    const int i = i0 + tid / kMaxK, j = tid % kMaxK;
    bool link = false;
    if (i < n1 && j < n2) {
      float acc = 0.f;
      for (int mm = 0; mm < M; ++mm) acc += r1[i * d.Mmax + mm] * r2[j * d.Mmax + mm];
      link = acc > 0.f;
    }
    // the 32 lanes that share i are contiguous (two rows per wave)
    const unsigned long long bal = __ballot(link);
    const unsigned word = (unsigned)(bal >> (((tid & 63) / kMaxK) * kMaxK));
    if (j == 0 && i < kMaxK) bonus[o * kMaxK + i] = i < n1 ? word : 0u;
  }
}

// grid (P), one wave.  LDS: back-pointers [T][kMaxK] u8, alive [T] u32, path [T] u8.
__global__ __launch_bounds__(64) void tube_viterbi_kernel(const float* __restrict__ kscore, const int* __restrict__ kn,
                                                          const unsigned* __restrict__ bonus, TubeDims d,
                                                          int* __restrict__ path_idx, float* __restrict__ path_total,
                                                          int* __restrict__ n_paths) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* back = smem;                                              // [T][kMaxK]
  unsigned* alive = reinterpret_cast<unsigned*>(smem + (size_t)d.T * kMaxK);   // [T]
  unsigned char* path = reinterpret_cast<unsigned char*>(alive + d.T);     // [T]
  const int p = blockIdx.x, lane = threadIdx.x, T = d.T;
  const float* S = kscore + (long)p * T * kMaxK;
  const unsigned* Bn = bonus + (long)p * T * kMaxK;
  const int* N = kn + (long)p * T;
  int K = kMaxK;
  for (int t = lane; t < T; t += 64) {
    const int n = N[t];
    alive[t] = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    K = min(K, n);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) K = min(K, __shfl_xor(K, off, 64));   // paths = min_t n_t (:161, :248-252)
  if (T <= 0) K = 0;
  if (lane == 0) n_paths[p] = K;
  __syncthreads();
  const float num_frames = (float)d.F;
  for (int k = 0; k < K; ++k) {
    // ---- forward pass, t = T-2 .. 0: D[t][a] = max_b ((s[t][a] + s[t+1][b]) (+1) + D[t+1][b]) over alive b (:207-218)
    float D = 0.f;                                        // data_scores[T-1] = 0
    // rows are fetched one step ahead of their use: the recursion is a chain of dependent steps, so an exposed global
    // load per step would be its whole cost
    float sb = (lane < kMaxK && T >= 2) ? S[(T - 1) * kMaxK + lane] : 0.f;      // scores of frame t + 1
    float sa = (lane < kMaxK && T >= 2) ? S[(T - 2) * kMaxK + lane] : 0.f;      // scores of frame t
    unsigned bn = (lane < kMaxK && T >= 2) ? Bn[(T - 2) * kMaxK + lane] : 0u;
    for (int t = T - 2; t >= 0; --t) {
      const int tp = max(t - 1, 0);
      const float sa_pref = lane < kMaxK ? S[tp * kMaxK + lane] : 0.f;
      const unsigned bn_pref = lane < kMaxK ? Bn[tp * kMaxK + lane] : 0u;
      unsigned al = __builtin_amdgcn_readfirstlane(alive[t + 1]);   // wave-uniform: the loop runs on the scalar unit
      float best = -INFINITY;
      int arg = 0;
      while (al) {
        const int b = __builtin_ctz(al);
        al &= al - 1;
        float e = sa + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sb), b));
        if ((bn >> b) & 1u) e += 1.0f;
        e = e + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D), b));
        if (e > best) { best = e; arg = b; }               // first maximum wins
      }
      if (lane < kMaxK) back[t * kMaxK + lane] = (unsigned char)arg;
      D = best;
      sb = sa; sa = sa_pref; bn = bn_pref;
    }
    // ---- start box: highest D[0] among the alive boxes of frame 0, lowest index among equals (:222-223)
    const unsigned a0 = alive[0];
    const bool mine = lane < kMaxK && ((a0 >> lane) & 1u);
    float v = mine ? D : -INFINITY;
    float vmax = v;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    const unsigned long long eq = __ballot(mine && v == vmax);
    int cur = __builtin_ctzll(eq);
    const float score = __shfl(D, cur, 64);
    __syncthreads();   // back[] complete
    if (lane == 0) {
      path[0] = (unsigned char)cur;
      for (int j = 0; j + 1 < T; ++j) {                   // backtrace (:228-232)
        cur = back[j * kMaxK + cur];
        path[j + 1] = (unsigned char)cur;
      }
      path_total[p * kMaxK + k] = score / num_frames;      // :233
    }
    __syncthreads();
    int* out = path_idx + ((long)p * kMaxK + k) * T;
    for (int t = lane; t < T; t += 64) {
      const int c = path[t];
      out[t] = c;
      alive[t] &= ~(1u << c);                              // :248-258
    }
    __syncthreads();
  }
}

int next_pow2_64(int v) { int p = 64; while (p < v) p <<= 1; return p; }

}  // namespace

extern "C" size_t dtt_tube_link_workspace_bytes(int problems, int frames) {
  if (problems <= 0 || frames < 2) return 0;
  return (size_t)problems * (frames - 1) * kMaxK * sizeof(unsigned);   // tracklet-link masks
}

extern "C" int dtt_tube_link(const float* dets, int det_stride, const int* n, const float* trk, const int* m,
                             int problems, int frames, int max_dets, int max_tracklets, int max_per_image,
                             float nms_thresh, float* kept_boxes, float* kept_scores, int* kept_n, int* path_idx,
                             float* path_total, int* n_paths, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(dets && n && kept_boxes && kept_scores && kept_n && path_idx && path_total && n_paths, "tube_link: null pointer");
  DTT_REQUIRE(problems > 0 && frames >= 2, "tube_link: need at least one problem and two frames");
  DTT_REQUIRE(det_stride >= 5, "tube_link: detection rows are [x1,y1,x2,y2,score,...]");
  DTT_REQUIRE(max_dets > 0 && max_dets <= kMaxDets, "tube_link: at most %d detections per frame (got %d)", kMaxDets, max_dets);
  DTT_REQUIRE(max_per_image > 0 && max_per_image <= kMaxK, "tube_link: max_per_image must be in 1..%d", kMaxK);
  DTT_REQUIRE(!trk || (m && max_tracklets > 0 && max_tracklets <= kMaxTrk), "tube_link: at most %d tracklets per frame", kMaxTrk);
  const int T = frames - 1;
  DTT_REQUIRE((size_t)T * (kMaxK + 5) + 64 <= 160 * 1024, "tube_link: too many frames (%d) for the LDS back-pointer table", frames);
  const size_t need = dtt_tube_link_workspace_bytes(problems, frames);
  DTT_REQUIRE(workspace && workspace_bytes >= need, "tube_link: workspace too small (%zu < %zu)", workspace_bytes, need);
  TubeDims d;
  d.P = problems; d.F = frames; d.T = T; d.Nmax = max_dets; d.Mmax = trk ? max_tracklets : 1; d.K = max_per_image;
  d.det_stride = det_stride;
  unsigned* bonus = static_cast<unsigned*>(workspace);
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();   // the attribute is per device, not per process
  if (!attr) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(tube_frame_nms_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(tube_bonus_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(tube_viterbi_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DTT_REQUIRE(e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess, "tube_link: cannot raise dynamic LDS limit");
    attr = true;
  }
  const int P2 = next_pow2_64(max_dets), W = P2 / 64;
  const size_t lds1 = (size_t)P2 * 16 + (size_t)P2 * W * 8 + (size_t)P2 * 2 + 16;
  hipLaunchKernelGGL(tube_frame_nms_kernel, dim3(T, problems), dim3(kThreads), lds1, stream, dets, n, d, nms_thresh, P2,
                     kept_boxes, kept_scores, kept_n);
  DTT_CHECK_LAUNCH("tube_frame_nms_kernel");
  if (T >= 2) {
    const size_t lds2 = (size_t)2 * kMaxK * d.Mmax * sizeof(float);
    hipLaunchKernelGGL(tube_bonus_kernel, dim3(T - 1, problems), dim3(kThreads), lds2, stream, kept_boxes, kept_n, trk, m,
                       d, bonus);
    DTT_CHECK_LAUNCH("tube_bonus_kernel");
  }
  const size_t lds3 = (size_t)T * kMaxK + (size_t)T * 4 + (size_t)T + 16;
  hipLaunchKernelGGL(tube_viterbi_kernel, dim3(problems), dim3(64), lds3, stream, kept_scores, kept_n, bonus, d,
                     path_idx, path_total, n_paths);
  DTT_CHECK_LAUNCH("tube_viterbi_kernel");
  return 1;
}

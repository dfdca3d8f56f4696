// RPN proposal layer for gfx950: one call for the whole batch, nothing leaves the device.
//
// Replaces _ProposalLayer.forward (reference rpn/proposal_layer.py:49-161): numpy meshgrid + H2D every
// call, decode/clip of all K*A anchors, a full torch.sort of the B x K*A scores, then a Python loop of
// per-image NMS round trips.  Here:
//   1. the selection: the pre_nms_topN best (score, index) keys of every image straight from the NCHW score map, in order.
//      Keys are (descending score, ascending anchor index), a total order, so the result is deterministic.
//        * proposal_sort_runs + proposal_rank_scatter (maps up to 38 912 anchors, i.e. every D&T shape): runs of 1024
//          consecutive anchor indices sorted in LDS by one workgroup each, then every key ranked against the other runs by
//          binary search with all sorted keys of the image staged in LDS -- the whole chip instead of one CU per image;
//        * proposal_select_sort (larger maps): one 1024-thread workgroup per image, exact bisection on the key bits
//          (keys cached in registers) + a bitonic sort of the selected keys in LDS.
//      Needs the SCORES only: callers that overlap the proposal layer with other work can start it as soon as the softmax
//      is done, while the box-delta convolution still runs (dtt_proposal_select_sort / dtt_proposal_decode_nms).
//   2. decode + clip of only the survivors (bbox_transform.py:108-134, 156-173): by the ranking kernel itself when the box
//      deltas are at hand (dtt_proposal_forward), else by proposal_decode (all CUs).
//   3. the batched NMS of nms.hip (mask tiles + on-device sweep) whose epilogue writes the zero-padded
//      (B, post_nms_topN, 5) RoI tensor (proposal_layer.py:151-159).
#include "common.h"
#include "nms_internal.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxSort = 16384;  // LDS-resident bitonic sort capacity (128 KiB of 64-bit keys)

__device__ __forceinline__ unsigned desc_key(float s) {
  s = s + 0.0f;  // -0.0 -> +0.0 so that both zeros tie, as a float comparison would
  unsigned u = __float_as_uint(s);
  unsigned asc = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
  return ~asc;  // smaller key = larger score
}

struct PropGeom {
  int A, H, W, K, n;       // anchors per cell, map size, cells, K*A
  int feat_stride;
  int topn;                // boxes handed to NMS per image
  int P;                   // power-of-two sort size >= topn
};

#ifdef DTT_WG_TRACE   // developer build: where and when the select / sort workgroups ran (tools/wg_trace.py)
__device__ unsigned long long dtt_sort_trace[16 * 8 * 8];
#define SORT_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 8) dtt_sort_trace[(trace_slot * 8 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
__device__ int dtt_sort_trace_launch;
#endif

#ifndef DTT_WG_TRACE
#define SORT_STAMP(k) do {} while (0)
#endif

constexpr int kEPT = 32;     // score keys cached in registers per thread (n <= 32768), else re-read from L2

// Block-wide sum of a per-wave (uniform) count: lane 0 of each wave posts it, everybody adds the 16 posts.
__device__ __forceinline__ unsigned block_sum(unsigned wave_count, unsigned* wsum) {
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) wsum[tid >> 6] = wave_count;
  __syncthreads();
  unsigned tot = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) tot += wsum[w];
  __syncthreads();
  return tot;
}

// Sort buffer index: one 8-byte pad after every 8 keys.  A thread of a sort pass owns 2 / 4 / 8 keys a power-of-two stride
// apart; with the pad, the 32 lanes of a ds_read_b64 fall in 32 different bank pairs for the stride-1 pass (72-byte lane
// stride) as well as for the wider ones.
__device__ __forceinline__ int sort_slot(int i) { return i + (i >> 3); }
__host__ __device__ constexpr int sort_slots(int P) { return P + (P >> 3); }

// NST consecutive stages (j = jlow << (NST-1), ..., 2*jlow, jlow) of bitonic phase k in ONE pass over LDS: a thread loads
// the 2^NST keys that only exchange among themselves in these stages, runs the compare-exchanges in registers and stores
// them back -- 35 barrier-separated passes for 8192 keys instead of 91.
template <int NST, int NT = kThreads>
__device__ __forceinline__ void bitonic_pass(unsigned long long* __restrict__ buf, int P, int k, int jlow, int tid) {
  constexpr int G = 1 << NST;
  const int plow = __builtin_ctz(jlow);
  for (int gi = tid; gi < (P >> NST); gi += NT) {
    const int base = ((gi >> plow) << (plow + NST)) | (gi & (jlow - 1));
    const bool up = (base & k) == 0;
    unsigned long long v[G];
#pragma unroll
    for (int e = 0; e < G; ++e) v[e] = buf[sort_slot(base + e * jlow)];
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      constexpr int dummy = 0; (void)dummy;
      const int bit = G >> (s + 1);
#pragma unroll
      for (int e = 0; e < G; ++e) {
        if (e & bit) continue;
        const unsigned long long lo = v[e], hi = v[e | bit];
        const bool sw = (lo > hi) == up;
        v[e] = sw ? hi : lo;
        v[e | bit] = sw ? lo : hi;
      }
    }
#pragma unroll
    for (int e = 0; e < G; ++e) buf[sort_slot(base + e * jlow)] = v[e];
  }
}

// LDS: buf[sort_slots(P)] u64 | wsum[2][16] (+ spare) | ctl[8]
// Slot e of thread tid is score-map element m = tid + e*kThreads in MEMORY order (m = a*K + k, coalesced);
// its flattened anchor index is t = k*A + a (proposal_layer.py:102-103).
//
// Selection of the topn best keys is an exact MSB-first bisection on the key bits: 32 rounds of
// "how many keys are below this pivot", each a compare + ballot + popcount per cached key and one
// 16-way sum through LDS (double-buffered: one barrier per round) -- no atomics, no histogram, insensitive to how
// skewed the scores are.  (Two bits per round -- three pivots, half the rounds -- was measured slower: the per-key
// s_bcnt1 / s_add pairs of all 16 waves go through the CU's one scalar unit, 53 us against 30.)
template <bool CACHE>
__global__ __launch_bounds__(kThreads) void proposal_select_sort(const float* __restrict__ cls_prob, PropGeom g,
                                                                 unsigned* __restrict__ order_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* buf = reinterpret_cast<unsigned long long*>(smem);
  unsigned* wsum = reinterpret_cast<unsigned*>(buf + sort_slots(g.P));
  unsigned* ctl = wsum + 96;  // [3] fill counter
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
#ifdef DTT_WG_TRACE
  const int trace_slot = dtt_sort_trace_launch % 16;
  if (tid == 0 && b < 8) {
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dtt_sort_trace[(trace_slot * 8 + b) * 8 + 0] = wall_clock64();
    dtt_sort_trace[(trace_slot * 8 + b) * 8 + 1] = ((unsigned long long)(xcc & 15) << 16) | (((id >> 13) & 7) << 8) | ((id >> 8) & 15);
  }
#endif
  const float* sc = cls_prob + ((long)b * 2 * g.A + g.A) * g.K;  // fg scores: channels A .. 2A-1
  const int nslots = (g.n + kThreads - 1) / kThreads;

  unsigned kc[kEPT];
  unsigned tc[kEPT];   // flattened anchor index t = k*A + a of slot e (m = a*K + k walked incrementally: no division per slot)
  if constexpr (CACHE) {
    int a = tid / g.K, k = tid - a * g.K;
    const int da = kThreads / g.K, dk = kThreads - da * g.K;
#pragma unroll
    for (int e = 0; e < kEPT; ++e) {
      const int m = tid + e * kThreads;
      kc[e] = m < g.n ? desc_key(sc[m]) : 0xFFFFFFFFu;  // padding is never below a pivot
      tc[e] = (unsigned)(k * g.A + a);
      k += dk; a += da;
      if (k >= g.K) { k -= g.K; ++a; }
    }
  }
  auto t_of = [&](int m) -> unsigned {
    const int a = m / g.K, k = m - a * g.K;
    return (unsigned)(k * g.A + a);
  };
  // number of keys strictly below `pivot`, over the whole image
  auto count_below = [&](unsigned pivot) -> unsigned {
    unsigned c = 0;
    if constexpr (CACHE) {
#pragma unroll
      for (int e = 0; e < kEPT; ++e) c += (unsigned)__builtin_popcountll(__ballot(kc[e] < pivot));
    } else {
      for (int e = 0; e < nslots; ++e) {
        const int m = tid + e * kThreads;
        c += (unsigned)__builtin_popcountll(__ballot(m < g.n && desc_key(sc[min(m, g.n - 1)]) < pivot));
      }
    }
    return block_sum(c, wsum);
  };

  SORT_STAMP(2);
  unsigned thr_hi = 0xFFFFFFFFu, thr_lo = 0xFFFFFFFFu;  // select (key32, t) <= (thr_hi, thr_lo)
  if (g.topn < g.n) {
    // largest T with count(key < T) < topn  ==  the topn-th smallest key
    unsigned T = 0;
    const int wave = tid >> 6;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned test = T | (1u << bit);
      unsigned c = 0;
      if constexpr (CACHE) {
#pragma unroll
        for (int e = 0; e < kEPT; ++e) c += (unsigned)__builtin_popcountll(__ballot(kc[e] < test));
      } else {
        for (int e = 0; e < nslots; ++e) {
          const int m = tid + e * kThreads;
          c += (unsigned)__builtin_popcountll(__ballot(m < g.n && desc_key(sc[min(m, g.n - 1)]) < test));
        }
      }
      unsigned* ws = wsum + (bit & 1) * 16;   // double-buffered: a slower wave may still be reading the previous round's sums
      if (lane == 0) ws[wave] = c;
      __syncthreads();                          // one barrier per round
      unsigned tot = 0;
#pragma unroll
      for (int w = 0; w < kThreads / 64; ++w) tot += ws[w];
      if (tot < (unsigned)g.topn) T = test;
    }
    __syncthreads();   // (the counts below reuse the first sum buffer)
    thr_hi = T;
    const unsigned below = count_below(T);
    const unsigned upto = T == 0xFFFFFFFFu ? (unsigned)g.n : count_below(T + 1u);
    const unsigned need = (unsigned)g.topn - below;  // how many of the keys tied at T are wanted
    if (upto - below != need) {
      // resolve the tie on the anchor index (lower index first): same bisection over t, tied keys only
      unsigned Tt = 0;
      int tb = 0;
      while ((1u << tb) < (unsigned)g.n) ++tb;
#pragma unroll 1
      for (int bit = tb - 1; bit >= 0; --bit) {
        const unsigned test = Tt | (1u << bit);
        unsigned c = 0;
        if constexpr (CACHE) {
#pragma unroll
          for (int e = 0; e < kEPT; ++e)   // (padding slots carry key 0xFFFFFFFF: equal to T only if they are wanted anyway ...
            c += (unsigned)__builtin_popcountll(__ballot(kc[e] == T && tid + e * kThreads < g.n && tc[e] < test));   // ... so test m < n)
        } else {
          for (int e = 0; e < nslots; ++e) {
            const int m = tid + e * kThreads;
            const int mm = min(m, g.n - 1);
            c += (unsigned)__builtin_popcountll(__ballot(m < g.n && desc_key(sc[mm]) == T && t_of(mm) < test));
          }
        }
        if (block_sum(c, wsum) < need) Tt = test;
      }
      thr_lo = Tt;
    }
  }
  SORT_STAMP(3);
  // ---- compact the selected keys into LDS (any order; one LDS atomic per wave), pad, sort
  if (tid == 0) ctl[3] = 0;
  __syncthreads();
  {
    auto push = [&](bool take, unsigned long long k64) {
      const unsigned long long mk = __ballot(take);
      if (mk) {
        unsigned basei = 0;
        const int leader = __builtin_ctzll(mk);
        if (lane == leader) basei = atomicAdd(&ctl[3], (unsigned)__builtin_popcountll(mk));
        basei = __builtin_amdgcn_readlane(basei, leader);
        if (take) buf[sort_slot((int)basei + __builtin_popcountll(mk & ((1ULL << lane) - 1ULL)))] = k64;
      }
    };
    if constexpr (CACHE) {
#pragma unroll
      for (int e = 0; e < kEPT; ++e) {
        const int m = tid + e * kThreads;
        const unsigned t = tc[e];
        const bool take = m < g.n && (kc[e] < thr_hi || (kc[e] == thr_hi && t <= thr_lo));
        push(take, ((unsigned long long)kc[e] << 32) | t);
      }
    } else {
      for (int e = 0; e < nslots; ++e) {
        const int m = tid + e * kThreads;
        const unsigned k32 = desc_key(sc[min(m, g.n - 1)]);
        const unsigned t = t_of(min(m, g.n - 1));
        const bool take = m < g.n && (k32 < thr_hi || (k32 == thr_hi && t <= thr_lo));
        push(take, ((unsigned long long)k32 << 32) | t);
      }
    }
  }
  for (int i = g.topn + tid; i < g.P; i += kThreads) buf[sort_slot(i)] = ~0ULL;
  __syncthreads();
  SORT_STAMP(4);
  for (int k = 2; k <= g.P; k <<= 1) {
    int j = k >> 1;
    while (j > 0) {   // stages j, j/2, j/4 (as many as the phase has left) per pass
      if (j >= 4) { bitonic_pass<3>(buf, g.P, k, j >> 2, tid); j >>= 3; }
      else if (j == 2) { bitonic_pass<2>(buf, g.P, k, 1, tid); j = 0; }
      else { bitonic_pass<1>(buf, g.P, k, 1, tid); j = 0; }
      __syncthreads();
    }
  }
  SORT_STAMP(5);
  // ---- rank r -> flattened anchor index t = k * A + a
  for (int r = tid; r < g.topn; r += kThreads) order_out[(long)b * g.topn + r] = (unsigned)buf[sort_slot(r)];
#ifdef DTT_WG_TRACE
  __syncthreads();
  if (tid == 0 && b < 8) dtt_sort_trace[(trace_slot * 8 + b) * 8 + 7] = wall_clock64();
  if (tid == 0 && b == 0) dtt_sort_trace_launch = dtt_sort_trace_launch + 1;   // (racy by design: approximate slotting is enough)
#endif
}

// Decode + clip one box (bbox_transform.py:108-134, 156-173): flattened anchor index t = k * A + a of image b.
__device__ __forceinline__ float4 decode_box(unsigned t, const float* __restrict__ dl, const float* __restrict__ anchors,
                                             const PropGeom& g, float xmax, float ymax) {
  const int k = t / g.A, a = t - k * g.A;
  const int h = k / g.W, w = k - h * g.W;
  const float sx = (float)(w * g.feat_stride), sy = (float)(h * g.feat_stride);
  const float x1 = anchors[a * 4 + 0] + sx, y1 = anchors[a * 4 + 1] + sy;
  const float x2 = anchors[a * 4 + 2] + sx, y2 = anchors[a * 4 + 3] + sy;
  const float dx = dl[(long)(4 * a + 0) * g.K + k], dy = dl[(long)(4 * a + 1) * g.K + k];
  const float dw = dl[(long)(4 * a + 2) * g.K + k], dh = dl[(long)(4 * a + 3) * g.K + k];
  const float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;
  const float ctr_x = x1 + 0.5f * widths, ctr_y = y1 + 0.5f * heights;
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = (float)exp((double)dw) * widths;   // correctly rounded binary32 exp (declared semantics)
  const float ph = (float)exp((double)dh) * heights;
  float4 o;
  o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), xmax);
  o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), ymax);
  o.z = fminf(fmaxf(pcx + 0.5f * pw, 0.f), xmax);
  o.w = fminf(fmaxf(pcy + 0.5f * ph, 0.f), ymax);
  return o;
}

// Decode + clip the survivors, one thread per (image, rank).
// Ranks [r0, r1) of every image; with `done` (the NMS's per-image flags, `done_stride` words apart) images whose NMS has
// finished are skipped (second half of a two-phase run).
__global__ __launch_bounds__(256) void proposal_decode(const unsigned* __restrict__ order, const float* __restrict__ bbox_pred,
                                                       const float* __restrict__ im_info, const float* __restrict__ anchors,
                                                       PropGeom g, float* __restrict__ boxes_out, int r0, int r1,
                                                       const unsigned long long* __restrict__ done, long done_stride) {
  const int b = blockIdx.y, r = r0 + blockIdx.x * 256 + threadIdx.x;
  if (r >= r1) return;
  if (done && done[b * done_stride] != 0ULL) return;
  const float im_h = im_info[b * 3 + 0], im_w = im_info[b * 3 + 1];
  const unsigned t = order[(long)b * g.topn + r];
  reinterpret_cast<float4*>(boxes_out)[(long)b * g.topn + r] =
      decode_box(t, bbox_pred + (long)b * 4 * g.A * g.K, anchors, g, im_w - 1.0f, im_h - 1.0f);
}

// ---- The selection on many workgroups (round 3): runs of 1024 keys sorted locally, then ranked against each other.
//
// The one-workgroup-per-image kernel above takes 85 us for 30552 anchors (32 bisection rounds + a 8192-key bitonic sort on 4
// CUs of 256).  Here the image's keys are cut into runs of kRun CONSECUTIVE FLATTENED ANCHOR INDICES t (so that the tie-break
// "lower t first" between two runs is just "lower run first", and within a run "lower position first"):
//   proposal_sort_runs     one 256-thread workgroup per run: keys (descending score) + position sorted in LDS, written as
//                          32-bit keys and the t they belong to;
//   proposal_rank_scatter  every workgroup stages ALL sorted runs' keys of its image in LDS (4 B x n <= 152 KB) and ranks two
//                          run: rank = own position + sum over the other runs of a binary search (count of keys <= mine in earlier
//                          runs, < mine in later runs) -- the exact position in the (score desc, t asc) order.  A lane stops
//                          as soon as its rank reaches topn (most anchors: the lanes of a wave hold neighbours of one sorted
//                          run, so they stop together).  Ranks below topn write order[rank] = t and, when the box deltas are
//                          at hand, the decoded box.
// No atomics, no cross-workgroup protocol: the second kernel only reads what the first wrote.
constexpr int kRun = 1024;
constexpr int kRunThreads = 256;
constexpr int kMaxRuns = 38;   // kMaxRuns * kRun * 4 B of LDS in proposal_rank_scatter

__global__ __launch_bounds__(kRunThreads) void proposal_sort_runs(const float* __restrict__ cls_prob, PropGeom g, int nruns,
                                                                  unsigned* __restrict__ rkeys, unsigned* __restrict__ rts) {
  __shared__ unsigned long long buf[sort_slots(kRun)];
  const int w = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* sc = cls_prob + ((long)b * 2 * g.A + g.A) * g.K;  // fg scores: channels A .. 2A-1
#pragma unroll
  for (int e = 0; e < kRun / kRunThreads; ++e) {
    const int i = tid + e * kRunThreads;
    const int t = w * kRun + i;
    unsigned key = 0xFFFFFFFFu;
    if (t < g.n) {
      const int k = t / g.A, a = t - k * g.A;
      key = desc_key(sc[(long)a * g.K + k]);
    }
    buf[sort_slot(i)] = ((unsigned long long)key << 32) | (unsigned)(t < g.n ? i : 0xFFFFFFFFu);
  }
  __syncthreads();
  for (int k = 2; k <= kRun; k <<= 1) {
    int j = k >> 1;
    while (j > 0) {
      if (j >= 4) { bitonic_pass<3, kRunThreads>(buf, kRun, k, j >> 2, tid); j >>= 3; }
      else if (j == 2) { bitonic_pass<2, kRunThreads>(buf, kRun, k, 1, tid); j = 0; }
      else { bitonic_pass<1, kRunThreads>(buf, kRun, k, 1, tid); j = 0; }
      __syncthreads();
    }
  }
  unsigned* ko = rkeys + ((long)b * nruns + w) * kRun;
  unsigned* to = rts + ((long)b * nruns + w) * kRun;
#pragma unroll
  for (int e = 0; e < kRun / kRunThreads; ++e) {
    const int i = tid + e * kRunThreads;
    const unsigned long long v = buf[sort_slot(i)];
    const unsigned lo = (unsigned)v;
    ko[i] = (unsigned)(v >> 32);
    to[i] = lo == 0xFFFFFFFFu ? 0xFFFFFFFFu : (unsigned)(w * kRun) + lo;
  }
}

constexpr int kRankThreads = 1024;  // one position of each of the workgroup's runs per thread
constexpr int kRunsPerWg = 2;       // few, fat workgroups.  Alone on the chip one run per workgroup is faster (16 us against 24), but
                                    // the layer runs BESIDE the conv4 correlation (two rounds of workgroups on 240 CUs): 60 waiting
                                    // rank workgroups take the CUs its second round needs (conv4 60 -> 72 us), 30 do not

constexpr int kRunsInFlight = 8;    // independent binary searches per thread (the search is a chain of dependent LDS reads)

template <bool DECODE>
__global__ __launch_bounds__(kRankThreads) void proposal_rank_scatter(const unsigned* __restrict__ rkeys, const unsigned* __restrict__ rts,
                                                                      PropGeom g, int nruns, unsigned* __restrict__ order,
                                                                      const float* __restrict__ bbox_pred,
                                                                      const float* __restrict__ im_info,
                                                                      const float* __restrict__ anchors, float* __restrict__ boxes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* lk = reinterpret_cast<unsigned*>(smem);   // [nruns][kRun]
  const int b = blockIdx.y, tid = threadIdx.x, pos = tid;
  {
    const uint4* src = reinterpret_cast<const uint4*>(rkeys + (long)b * nruns * kRun);
    uint4* dst = reinterpret_cast<uint4*>(lk);
    const int total = nruns * (kRun / 4);
    for (int i0 = tid; i0 < total; i0 += 8 * kRankThreads) {   // 8 loads in flight per thread: the whole image in one round trip
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + u * kRankThreads, total - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * kRankThreads < total) dst[i0 + u * kRankThreads] = v[u];
    }
  }
  __syncthreads();
  int w[kRunsPerWg], rank[kRunsPerWg];
  unsigned key[kRunsPerWg];
#pragma unroll
  for (int q = 0; q < kRunsPerWg; ++q) {
    w[q] = blockIdx.x * kRunsPerWg + q;
    const bool have = w[q] < nruns;
    key[q] = have ? lk[w[q] * kRun + pos] : 0u;
    rank[q] = have ? 0 : g.topn;   // (a run past the end: never live)
  }
#pragma unroll 1
  for (int r0 = 0; r0 < nruns; r0 += kRunsInFlight) {
    bool live = false;
#pragma unroll
    for (int q = 0; q < kRunsPerWg; ++q) live = live || rank[q] < g.topn;
    if (!__any(live)) break;
#pragma unroll
    for (int q = 0; q < kRunsPerWg; ++q) {
      // count of keys that precede mine in runs r0 .. r0+7: x <= key in an earlier run, x < key in a later one, my own
      // position in my own run
      const unsigned* a[kRunsInFlight];
      bool le[kRunsInFlight];
      int lo[kRunsInFlight];
#pragma unroll
      for (int u = 0; u < kRunsInFlight; ++u) {
        const int r = min(r0 + u, nruns - 1);
        a[u] = lk + r * kRun;
        le[u] = r < w[q];
        lo[u] = 0;
      }
#pragma unroll
      for (int s = kRun / 2; s > 0; s >>= 1) {
        unsigned x[kRunsInFlight];
#pragma unroll
        for (int u = 0; u < kRunsInFlight; ++u) x[u] = a[u][lo[u] + s - 1];
#pragma unroll
        for (int u = 0; u < kRunsInFlight; ++u) lo[u] += (x[u] < key[q] || (le[u] && x[u] == key[q])) ? s : 0;
      }
      {
        unsigned x[kRunsInFlight];
#pragma unroll
        for (int u = 0; u < kRunsInFlight; ++u) x[u] = a[u][lo[u]];
#pragma unroll
        for (int u = 0; u < kRunsInFlight; ++u) lo[u] += (x[u] < key[q] || (le[u] && x[u] == key[q])) ? 1 : 0;
      }
#pragma unroll
      for (int u = 0; u < kRunsInFlight; ++u) {
        const int r = r0 + u;
        rank[q] += r >= nruns ? 0 : (r == w[q] ? pos : lo[u]);
      }
    }
  }
  float xmax = 0.f, ymax = 0.f;
  if constexpr (DECODE) { ymax = im_info[b * 3 + 0] - 1.0f; xmax = im_info[b * 3 + 1] - 1.0f; }
#pragma unroll
  for (int q = 0; q < kRunsPerWg; ++q) {
    if (rank[q] >= g.topn) continue;
    const unsigned t = rts[((long)b * nruns + w[q]) * kRun + pos];
    if (t == 0xFFFFFFFFu) continue;   // padding of the last run
    order[(long)b * g.topn + rank[q]] = t;
    if constexpr (DECODE)
      reinterpret_cast<float4*>(boxes)[(long)b * g.topn + rank[q]] =
          decode_box(t, bbox_pred + (long)b * 4 * g.A * g.K, anchors, g, xmax, ymax);
  }
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int effective_topn(int batch, int n, int pre_nms_topN) {
  // proposal_layer.py:138-139: the guard compares against numel of the whole batch
  if (pre_nms_topN > 0 && (long)pre_nms_topN < (long)batch * n) return pre_nms_topN < n ? pre_nms_topN : n;
  return n;
}

}  // namespace

#ifdef DTT_WG_TRACE
extern "C" int dtt_sort_trace_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_sort_trace), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

namespace {

// workspace: boxes (B, topn, 4) | NMS mask | keep | num | order (B, topn)
struct PropPlan {
  PropGeom g;
  size_t off_mask, off_keep, off_num, off_order, off_rkeys, off_rts, total, mask_per_image;
  int nruns;   // > 0: the selection runs on many workgroups (proposal_sort_runs + proposal_rank_scatter)
};

PropPlan plan_proposal(int batch, int num_anchors, int height, int width, int feat_stride, int pre_nms_topN) {
  PropPlan p;
  PropGeom& g = p.g;
  g.A = num_anchors; g.H = height; g.W = width; g.K = height * width; g.n = g.K * g.A;
  g.feat_stride = feat_stride;
  g.topn = effective_topn(batch, g.n, pre_nms_topN);
  g.P = next_pow2(g.topn);
  p.mask_per_image = dtt_nms_mask_bytes(g.topn);
  p.off_mask = align_up((size_t)batch * g.topn * 4 * sizeof(float), 256);
  p.off_keep = p.off_mask + align_up((size_t)batch * p.mask_per_image, 256);
  p.off_num = p.off_keep + align_up((size_t)batch * g.topn * sizeof(int), 256);
  p.off_order = p.off_num + 256;
  p.total = p.off_order + align_up((size_t)batch * g.topn * sizeof(unsigned), 256);
  static const bool one_wg = getenv("DTT_PROPOSAL_ONE_WG") != nullptr;   // developer A/B switch: the one-workgroup-per-image selection
  const int nruns = (g.n + kRun - 1) / kRun;
  p.nruns = (nruns <= kMaxRuns && g.P <= kMaxSort && !one_wg) ? nruns : 0;
  p.off_rkeys = p.total;
  p.off_rts = p.off_rkeys + align_up((size_t)batch * nruns * kRun * sizeof(unsigned), 256);
  p.total = p.off_rts + align_up((size_t)batch * nruns * kRun * sizeof(unsigned), 256);
  return p;
}

}  // namespace

extern "C" size_t dtt_proposal_workspace_bytes(int batch, int num_anchors, int height, int width,
                                               int pre_nms_topN) {
  return plan_proposal(batch, num_anchors, height, width, 1, pre_nms_topN).total;
}

// The selection on many workgroups; with bbox_pred also decodes the selected boxes into the workspace.
static int select_runs(const PropPlan& p, const float* cls_prob, const float* bbox_pred, const float* im_info, const float* anchors,
                       int batch, unsigned char* w, hipStream_t stream) {
  const PropGeom& g = p.g;
  unsigned* order = reinterpret_cast<unsigned*>(w + p.off_order);
  unsigned* rkeys = reinterpret_cast<unsigned*>(w + p.off_rkeys);
  unsigned* rts = reinterpret_cast<unsigned*>(w + p.off_rts);
  float* boxes = reinterpret_cast<float*>(w);
  const size_t lds = (size_t)p.nruns * kRun * sizeof(unsigned);
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_rank_scatter<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_rank_scatter<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DTT_REQUIRE(e == hipSuccess && e2 == hipSuccess, "proposal: cannot raise dynamic LDS limit");
    attr = true;
  }
  dtt_prof_begin("proposal_select_sort", stream);
  hipLaunchKernelGGL(proposal_sort_runs, dim3(p.nruns, batch), dim3(kRunThreads), 0, stream, cls_prob, g, p.nruns, rkeys, rts);
  DTT_CHECK_LAUNCH("proposal_sort_runs");
  if (bbox_pred)
    hipLaunchKernelGGL(proposal_rank_scatter<true>, dim3((p.nruns + kRunsPerWg - 1) / kRunsPerWg, batch), dim3(kRankThreads), lds, stream, rkeys, rts, g, p.nruns,
                       order, bbox_pred, im_info, anchors, boxes);
  else
    hipLaunchKernelGGL(proposal_rank_scatter<false>, dim3((p.nruns + kRunsPerWg - 1) / kRunsPerWg, batch), dim3(kRankThreads), lds, stream, rkeys, rts, g, p.nruns,
                       order, nullptr, nullptr, nullptr, nullptr);
  dtt_prof_end("proposal_select_sort", stream);
  DTT_CHECK_LAUNCH("proposal_rank_scatter");
  return 1;
}

// Phase 1 of the proposal layer: per image, the pre_nms_topN best (score, anchor) keys in order -> workspace.  Reads the
// scores only.
extern "C" int dtt_proposal_select_sort(const float* cls_prob, int batch, int num_anchors, int height, int width,
                                        int pre_nms_topN, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(cls_prob, "proposal: null pointer");
  DTT_REQUIRE(batch > 0 && num_anchors > 0 && height > 0 && width > 0, "proposal: bad shape");
  const PropPlan p = plan_proposal(batch, num_anchors, height, width, 1, pre_nms_topN);
  const PropGeom& g = p.g;
  DTT_REQUIRE(g.P <= kMaxSort, "proposal: %d boxes per image exceed the %d-entry LDS sort", g.topn, kMaxSort);
  DTT_REQUIRE(workspace && workspace_bytes >= p.total, "proposal: workspace too small (%zu < %zu)", workspace_bytes, p.total);
  DTT_REQUIRE(batch <= 65535, "proposal: more than 65535 images in one call");
  if (p.nruns) return select_runs(p, cls_prob, nullptr, nullptr, nullptr, batch, static_cast<unsigned char*>(workspace), stream);
  unsigned* order = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(workspace) + p.off_order);
  const size_t lds = (size_t)sort_slots(g.P) * 8 + (96 + 8) * 4;
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();   // the attribute is per device, not per process
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_select_sort<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(proposal_select_sort<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DTT_REQUIRE(e == hipSuccess && e2 == hipSuccess, "proposal: cannot raise dynamic LDS limit");
    attr = true;
  }
  dtt_prof_begin("proposal_select_sort", stream);
  if (g.n <= kEPT * kThreads)
    hipLaunchKernelGGL(proposal_select_sort<true>, dim3(batch), dim3(kThreads), lds, stream, cls_prob, g, order);
  else
    hipLaunchKernelGGL(proposal_select_sort<false>, dim3(batch), dim3(kThreads), lds, stream, cls_prob, g, order);
  dtt_prof_end("proposal_select_sort", stream);
  DTT_CHECK_LAUNCH("proposal_select_sort");
  return 1;
}

// Phase 2: decode + clip the boxes phase 1 selected (same geometry arguments, same workspace), NMS, RoI tensor.
static int decode_nms(const float* bbox_pred, const float* im_info, const float* anchors, int batch,
                      int num_anchors, int height, int width, int feat_stride, int pre_nms_topN,
                      int post_nms_topN, float nms_thresh, float* rois_out, int* num_out, void* workspace,
                      size_t workspace_bytes, void* stream_, bool boxes_ready) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(bbox_pred && im_info && anchors && rois_out, "proposal: null pointer");
  DTT_REQUIRE(batch > 0 && num_anchors > 0 && height > 0 && width > 0 && feat_stride > 0, "proposal: bad shape");
  DTT_REQUIRE(post_nms_topN > 0, "proposal: post_nms_topN must be > 0");
  const PropPlan p = plan_proposal(batch, num_anchors, height, width, feat_stride, pre_nms_topN);
  const PropGeom& g = p.g;
  DTT_REQUIRE(workspace && workspace_bytes >= p.total, "proposal: workspace too small (%zu < %zu)", workspace_bytes, p.total);
  unsigned char* w = static_cast<unsigned char*>(workspace);
  float* boxes = reinterpret_cast<float*>(w);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(w + p.off_mask);
  int* keep = reinterpret_cast<int*>(w + p.off_keep);
  int* num_ws = reinterpret_cast<int*>(w + p.off_num);
  const unsigned* order = reinterpret_cast<const unsigned*>(w + p.off_order);
  DTT_REQUIRE(batch <= 65535, "proposal: more than 65535 images in one call");
  // Two halves when the NMS runs in two phases (nms.hip: post_nms_topN << topn): its first phase only looks at the first
  // `split` super-chunks of 1024 boxes, so only those are decoded in front of it; the rest is decoded (for the images that
  // still need it) between the phases.
  const long mask_stride = (long)(p.mask_per_image / sizeof(unsigned long long));
  const int split = dtt_nms_split(g.topn, post_nms_topN, 1);
  const int first = split ? min(g.topn, split * 1024) : g.topn;
  if (!boxes_ready) {
    hipLaunchKernelGGL(proposal_decode, dim3((first + 255) / 256, batch), dim3(256), 0, stream, order, bbox_pred, im_info, anchors, g,
                       boxes, 0, first, nullptr, 0L);
    DTT_CHECK_LAUNCH("proposal_decode");
  }
  int* nout = num_out ? num_out : num_ws;
  // (event tag nms_op: the layer's whole NMS -- both phases' mask and sweep launches and the decode of the second half between them)
  dtt_prof_begin("nms_op", stream);
  if (!dtt_nms_phase1(boxes, 4, (long)g.topn * 4, nullptr, g.topn, batch, nms_thresh, post_nms_topN, mask, mask_stride, keep, g.topn,
                      nout, rois_out, post_nms_topN, split, stream))
    return 0;
  if (!split) { dtt_prof_end("nms_op", stream); return 1; }
  if (g.topn > first && !boxes_ready) {
    const long cbw = (g.topn + 63) / 64;
    hipLaunchKernelGGL(proposal_decode, dim3((g.topn - first + 255) / 256, batch), dim3(256), 0, stream, order, bbox_pred, im_info,
                       anchors, g, boxes, first, g.topn, mask + (long)g.topn * cbw, mask_stride);
    DTT_CHECK_LAUNCH("proposal_decode (second half)");
  }
  const int ok = dtt_nms_phase2(boxes, 4, (long)g.topn * 4, nullptr, g.topn, batch, nms_thresh, post_nms_topN, mask, mask_stride, keep, g.topn,
                                nout, rois_out, post_nms_topN, split, stream);
  dtt_prof_end("nms_op", stream);
  return ok;
}

extern "C" int dtt_proposal_decode_nms(const float* bbox_pred, const float* im_info, const float* anchors, int batch,
                                       int num_anchors, int height, int width, int feat_stride, int pre_nms_topN,
                                       int post_nms_topN, float nms_thresh, float* rois_out, int* num_out, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
  return decode_nms(bbox_pred, im_info, anchors, batch, num_anchors, height, width, feat_stride, pre_nms_topN, post_nms_topN,
                    nms_thresh, rois_out, num_out, workspace, workspace_bytes, stream_, false);
}

extern "C" int dtt_proposal_forward(const float* cls_prob, const float* bbox_pred, const float* im_info,
                                    const float* anchors, int batch, int num_anchors, int height, int width,
                                    int feat_stride, int pre_nms_topN, int post_nms_topN, float nms_thresh,
                                    float* rois_out, int* num_out, void* workspace, size_t workspace_bytes,
                                    void* stream_) {
  DTT_REQUIRE(cls_prob && bbox_pred && im_info && anchors && rois_out, "proposal: null pointer");
  DTT_REQUIRE(post_nms_topN > 0, "proposal: post_nms_topN must be > 0");
  DTT_REQUIRE(batch > 0 && num_anchors > 0 && height > 0 && width > 0 && feat_stride > 0, "proposal: bad shape");
  const PropPlan p = plan_proposal(batch, num_anchors, height, width, feat_stride, pre_nms_topN);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int ok;
  dtt_prof_begin("proposal_op", stream);   // (event tag: the whole layer -- ranking, decode, NMS, RoI tensor)
  if (p.nruns) {
    // scores and box deltas both at hand: the ranking kernel decodes the boxes it selects (no decode launches)
    DTT_REQUIRE(workspace && workspace_bytes >= p.total, "proposal: workspace too small (%zu < %zu)", workspace_bytes, p.total);
    DTT_REQUIRE(batch <= 65535, "proposal: more than 65535 images in one call");
    if (!select_runs(p, cls_prob, bbox_pred, im_info, anchors, batch, static_cast<unsigned char*>(workspace), stream))
      return 0;
    ok = decode_nms(bbox_pred, im_info, anchors, batch, num_anchors, height, width, feat_stride, pre_nms_topN, post_nms_topN,
                    nms_thresh, rois_out, num_out, workspace, workspace_bytes, stream_, true);
  } else {
    if (!dtt_proposal_select_sort(cls_prob, batch, num_anchors, height, width, pre_nms_topN, workspace, workspace_bytes, stream_))
      return 0;
    ok = dtt_proposal_decode_nms(bbox_pred, im_info, anchors, batch, num_anchors, height, width, feat_stride, pre_nms_topN,
                                 post_nms_topN, nms_thresh, rois_out, num_out, workspace, workspace_bytes, stream_);
  }
  dtt_prof_end("proposal_op", stream);
  return ok;
}

// Gradient of the position-major PSRoI pooling + vote (heads.hip: psroi_pm_kernel) with respect to the map, for ALL the heads
// pooled from one map in one launch (gfx950).
//
// Reference: PSROIPoolBackward (psroi_pooling_kernel.cu:109-170) composed with the AvgPool2d vote (rfcn.py:62-64):
//     d map[b, h, w, bin, c] = sum over the RoIs r of image b whose bin `bin` contains (h, w) of  gvote[r, c] / (P*P) / area(r, bin)
// The reference scatters with atomicAdd, one launch per head.  Here the map is stationary and one WAVE owns a pixel:
//
//   prologue (once per workgroup, the only __syncthreads): the run of RoI rows that holds the image's RoIs is found, their
//             vote-gradient rows (all heads side by side, 36 floats per RoI for the class + box heads) are staged in LDS, their bin
//             edges computed with the forward's arithmetic (psroi_bin.h) and packed start | end << 16 -- no edges kernel, no
//             scratch, no memset in front of the launch;
//   phase 1   lanes = RoIs: the bins of lane's RoI that contain the pixel are a rectangle [ph_lo, ph_hi] x [pw_lo, pw_hi]
//             (edges are non-decreasing: the count of starts <= h and of ends <= h give the range with 14 compares); the row range
//             is recomputed only when the wave's pixel run enters a new map row, the column edges stay in registers;
//             an in-wave prefix sum (DPP) places every lane's (RoI, bin, weight) entries in RoI order in the wave's LDS list;
//   phase 2   lanes = columns (classes of head 0, then the box deltas of head 1): the list is walked in order, each entry one
//             read-modify-write of the wave's accumulator row in LDS (batches of 4 whose bins differ are issued together);
//   write-out the pixel's whole row (all bins of all heads, the padding columns as zeros, optionally plus a compact gradient that
//             autograd would otherwise add in a further pass) leaves as 16-byte stores and the accumulator is zeroed behind them.
//
// No atomics, no pre-zeroed output, summation in RoI order: run-to-run bit-identical.  The one-workgroup-per-pixel kernel this
// replaces (heads.hip: psroi_pm_bwd_kernel, kept behind DTT_PSROI_BWD_OLD=1 for the A/B) re-read the edges and the gradient rows
// from L2 for every pixel and took three workgroup barriers per pixel: 68.6 us for the class head + 12 us for the box head of the
// training step's 10184 pixels (profiles/r05_train_steady_state.txt).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "psroi_bin.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kListCap = 128;     // (RoI, bin, weight) entries of one wave between two walks
constexpr int kMaxRounds = 4;     // RoIs staged per chunk <= 64 * kMaxRounds
constexpr int kWOut = 4;          // 16-byte pieces of the accumulator row a lane has in flight in the write-out
constexpr int kWTab = 256;        // 1 / (P*P) / area for bin areas below this from a table (a bin is about 1/P of the map per side)

#ifdef DTT_PSROI_BWD_STAMP   // developer timeline (tools/psroi_bwd_timeline.py): shader-clock stamps of waves 0 and NW-1 of two workgroups
__device__ unsigned long long dtt_psroi_bwd_stamps[2 * 2 * 64];
__device__ unsigned long long dtt_psroi_bwd_wg[1024 * 3];      // per workgroup: entry, end (100 MHz clock), HW_ID
__device__ unsigned int dtt_psroi_bwd_cnt[1024 * 4];           // per workgroup: list entries walked, serial rounds, flushes, chunks
#define PB_STAMP(idx) do { const int sb_ = blockIdx.x == 0 ? 0 : blockIdx.x == gridDim.x / 2 ? 1 : -1; \
    if (sb_ >= 0 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == NW - 1) && (idx) < 64) \
      dtt_psroi_bwd_stamps[(sb_ * 2 + ((threadIdx.x >> 6) != 0)) * 64 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define PB_STAMP(idx) do {} while (0)
#endif

struct PmBwd {
  const float* gv0; const float* gv1;     // (num_rois, od0) / (num_rois, od1) or NULL
  int od0, cp0, od1, cp1;
  const float* rois; int num_rois; float spatial_scale; int batch_size;
  int height, width; long pixel_stride; int row_floats;   // columns written per pixel: [0, row_floats)
  const float* add; int add_first, add_count;             // optional (pixels, add_count) rows added into columns [add_first, +add_count)
  float* gmap;
  int cap;                                 // RoIs staged per chunk (multiple of 64, <= 64 * kMaxRounds)
  int per_image;                           // ceil(num_rois / batch_size): where image b's RoIs start when every image lists as many
  int ppw;                                 // pixels per wave
  int wgs_per_image;
  unsigned width_magic, wpi_magic;         // 2^32 / width + 1, 2^32 / wgs_per_image + 1 (0: divide)
  int ablate;                              // developer timing experiments (DTT_PSROI_BWD_ABLATE): 1 no listing, 2 no list walk, 4 no stores, 8 prologue only
};

// inclusive prefix sum over the 64 lanes (all active): row_shr 1 2 4 8 inside the rows of 16, then row_bcast 15 / 31
__device__ __forceinline__ int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}
// n / d through magic = 2^32 / d + 1 (exact while n * d < 2^32: the launcher passes 0 otherwise, and the division is done)
__device__ __forceinline__ int mdiv(int n, int d, unsigned magic) {
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n / d;
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sgprf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

template <int P, int NW, int NR>
__global__ __launch_bounds__(NW * 64) void psroi_pm_bwd_rows_kernel(PmBwd a) {
  static_assert(P <= 15 && P * P <= kListCap, "bin counts are packed in 4 bits; a lane's entries fit the list");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  PB_STAMP(0);
#ifdef DTT_PSROI_BWD_STAMP
  if (threadIdx.x < 4 && blockIdx.x < 1024) dtt_psroi_bwd_cnt[blockIdx.x * 4 + threadIdx.x] = 0;
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    dtt_psroi_bwd_wg[blockIdx.x * 3] = __builtin_amdgcn_s_memrealtime();
    dtt_psroi_bwd_wg[blockIdx.x * 3 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 11) | (0 << 6) | 4) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  }
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) dtt_psroi_bwd_stamps[(blockIdx.x == 0 ? 0 : 2) * 64 + 62] = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gvw = a.cp0 + a.cp1;
  const int cap = a.cap;
  // LDS: [per wave: accumulator row | list] [gradient rows cap x gvw] [row edges P x cap] [column edges P x cap] [non-zero flags cap] [own-image flags cap] [weights] [run]
  const int acc_floats = (a.row_floats + 3) & ~3;
  float* acc = smem + (long)wave * (acc_floats + 2 * kListCap + 4);
  int* list_rb = reinterpret_cast<int*>(acc + acc_floats);
  float* list_w = acc + acc_floats + kListCap;
  float* gvs = smem + (long)NW * (acc_floats + 2 * kListCap + 4);
  int* erow = reinterpret_cast<int*>(gvs + (long)cap * gvw);
  int* ecol = erow + P * cap;
  int* eimg = ecol + P * cap;
  int* eown = eimg + cap;                                 // [cap]: the RoI row belongs to this workgroup's image
  float* wtab = reinterpret_cast<float*>(eown + cap);
  int* run = reinterpret_cast<int*>(wtab + kWTab);       // [0 .. 1]: the image's run of RoI rows; [2]: the workgroup's pixel ticket

  const int b = mdiv(blockIdx.x, a.wgs_per_image, a.wpi_magic);
  const int hw = a.height * a.width;
  // Pixels are dealt ROUND ROBIN, consecutive pixels to DIFFERENT workgroups: pixel p of an image goes to workgroup p mod G, wave
  // (p / G) mod NW, round p / (G NW)  (G = wgs_per_image).  The RoIs crowd parts of the map: with a run of consecutive pixels per wave,
  // workgroup lifetimes were 14.7 .. 34.1 us (median 23.8); with 16 consecutive pixels per workgroup and round, the training step's
  // sampled RoIs (foreground around the ground truth) still left 1248 (median) .. 4393 list entries per workgroup.  Every workgroup now
  // samples the whole map.
  const int p_first = (blockIdx.x - b * a.wgs_per_image) + a.wgs_per_image * wave;
  const int p_step = a.wgs_per_image * NW;
  const float inv_bins = 1.f / (float)(P * P);
  const bool col_on = lane < gvw;

  // stage RoI rows [c0, c0 + n): gradient rows (all heads side by side), a flag per RoI whose rows are not all zeros (pre-zeroed), whether
  // the row belongs to this image, bin edges by the forward's arithmetic
  auto stage = [&](int c0, int n) {
    // a wave per RoI row, lane = column: the loads of a wave's rows are all in flight together (no index arithmetic per element)
    for (int r0 = wave; r0 < n; r0 += 4 * NW) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = min(r0 + u * NW, n - 1);
        v[u] = 0.f;
        if (lane < a.cp0) { if (lane < a.od0) v[u] = a.gv0[(long)(c0 + r) * a.od0 + lane]; }
        else if (lane - a.cp0 < a.od1) v[u] = a.gv1[(long)(c0 + r) * a.od1 + (lane - a.cp0)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * NW;
        if (r < n && col_on) gvs[r * gvw + lane] = v[u];
        if (r < n && v[u] != 0.f) eimg[r] = 1;       // (benign race: every writer stores 1)
      }
    }
    // a wave per (bin index k, 64 RoIs)
    const int nrd = (n + 63) >> 6;
    for (int t = wave, k = 0, rd = wave; t < P * nrd; t += NW, rd += NW) {
      while (rd >= nrd) { rd -= nrd; ++k; }
      const int r = rd * 64 + lane;
      if (r < n) {
        float roi[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) roi[q] = a.rois[(long)(c0 + r) * 5 + q];
        const Bin bn = psroi_bin(roi, a.spatial_scale, k, k, P, P, a.height, a.width);   // rows depend on ph only, columns on pw only
        erow[k * cap + r] = bn.hstart | (bn.hend << 16);
        ecol[k * cap + r] = bn.wstart | (bn.wend << 16);
        if (k == 0) eown[r] = min(max((int)roi[0], 0), a.batch_size - 1) == b ? 1 : 0;      // (the forward's clamp)
      }
    }
  };
  const int step_h = mdiv(p_step, a.width, a.width_magic), step_w = p_step - step_h * a.width;
  // ---- the run [r_first, r_end) of RoI rows that holds this image's RoIs (callers list their RoIs image by image: the run is the
  //      image's own RoIs; rows of other images inside it are masked), the weight table, the zeroed accumulator.  Callers that list
  //      the same number of RoIs per image (training) have image b's run at b * per_image: that chunk is staged WHILE the run is
  //      being looked for (one global-memory round trip less in front of the first pixel) and kept when the run lies inside it.
  if (tid < 3) run[tid] = tid == 0 ? a.num_rois : tid == 1 ? 0 : NW;      // (tickets 0 .. NW - 1: every wave's first pixel)
  for (int i = tid; i < kWTab; i += NW * 64) wtab[i] = inv_bins / (float)i;
  for (int i = tid; i < cap; i += NW * 64) eimg[i] = 0;
  for (int i = lane; i < acc_floats / 4; i += 64) reinterpret_cast<f32x4*>(acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  PB_STAMP(1);
  const int guess_c0 = min(b * a.per_image, a.num_rois), guess_n = min(cap, a.num_rois - guess_c0);
  {
    // (one LDS atomic pair per WAVE, from the ballot of its lanes' rows: 64 lanes on one address took 10 k cycles in front of barrier 2)
    int lo = a.num_rois, hi = 0;
    for (int r0 = wave * 64; r0 < a.num_rois; r0 += NW * 64) {
      const int r = r0 + lane;
      const bool own = r < a.num_rois && min(max((int)a.rois[(long)min(r, a.num_rois - 1) * 5], 0), a.batch_size - 1) == b;   // (the forward's clamp)
      const unsigned long long m = __ballot(own);
      if (m) { lo = min(lo, r0 + (int)__builtin_ctzll(m)); hi = max(hi, r0 + 64 - (int)__builtin_clzll(m)); }
    }
    if (guess_n > 0) stage(guess_c0, guess_n);
    if (hi > 0 && lane == 0) { atomicMin(&run[0], lo); atomicMax(&run[1], hi); }
  }
  PB_STAMP(2);
  __syncthreads();
  PB_STAMP(3);
  int r_first = run[0], r_end = run[1];
  const bool guessed = r_end > r_first && guess_n > 0 && r_first >= guess_c0 && r_end <= guess_c0 + guess_n;
  if (guessed) { r_first = guess_c0; r_end = guess_c0 + guess_n; }     // (the staged chunk, one chunk)
  const int nchunks = guessed ? 1 : r_end > r_first ? (r_end - r_first + cap - 1) / cap : 0;

  // lane = column of the accumulator row in phase 2
  // (lanes past the heads' columns add into a spare word behind the wave's list: straight-line code instead of four exec-mask regions per batch)
  char* const acc_lane = reinterpret_cast<char*>(!col_on ? acc + acc_floats + 2 * kListCap : acc + (lane < a.cp0 ? lane : P * P * a.cp0 + (lane - a.cp0)));
  const unsigned col_mul4 = !col_on ? 0u : 4u * (lane < a.cp0 ? a.cp0 : a.cp1);
  const float* const gv_lane = gvs + (col_on ? lane : 0);

  int cw[NR][P];                  // column edges of lane's RoIs (packed), per round of 64 RoIs
  int prow[NR];                   // row range of lane's RoIs at the current map row: lo | hi1 << 4 (hi1 = hi + 1; lo >= hi1: none)
  int cur_h = -1;

  // The list in order (= RoI order): each lane takes ONE entry of the next 64 into registers (one LDS read for 64 entries; a broadcast
  // read per entry cost as many LDS cycles as the accumulation itself) and the entries come back one by one through v_readlane;
  // accumulator[bin][lane's column] += gradient row[lane's column] * weight, four entries at a time -- read and written together when
  // their bins differ.  (ds_add_f32 instead of read - add - write keeps the bits and the order and needs no round trip, but the LDS
  // runs a float atomic at about 1000 cycles per instruction: 48 -> 191 us for the launch.)
  auto add1 = [&](int s, float sw, float g) {
    *reinterpret_cast<float*>(acc_lane + __umul24((unsigned)(s >> 8), col_mul4)) += g * sw;
  };
  auto walk = [&](int nl) {
#ifdef DTT_PSROI_BWD_STAMP
    if (lane == 0 && blockIdx.x < 1024) { atomicAdd(&dtt_psroi_bwd_cnt[blockIdx.x * 4], (unsigned)nl); atomicAdd(&dtt_psroi_bwd_cnt[blockIdx.x * 4 + 2], 1u); }
#endif
#pragma unroll 1
    for (int base = 0; base < nl; base += 64) {
      const int mine = min(base + lane, nl - 1);
      const int e = list_rb[mine];
      const int ew = __float_as_int(list_w[mine]);
      const int cnt = min(64, nl - base);
      int j = 0;
#pragma unroll 1
      for (; j + 4 <= cnt; j += 4) {
        const int s0 = __builtin_amdgcn_readlane(e, j), s1 = __builtin_amdgcn_readlane(e, j + 1);
        const int s2 = __builtin_amdgcn_readlane(e, j + 2), s3 = __builtin_amdgcn_readlane(e, j + 3);
        const float g0 = gv_lane[(s0 & 0xff) * gvw], g1 = gv_lane[(s1 & 0xff) * gvw];
        const float g2 = gv_lane[(s2 & 0xff) * gvw], g3 = gv_lane[(s3 & 0xff) * gvw];
        const float w0 = __int_as_float(__builtin_amdgcn_readlane(ew, j)), w1 = __int_as_float(__builtin_amdgcn_readlane(ew, j + 1));
        const float w2 = __int_as_float(__builtin_amdgcn_readlane(ew, j + 2)), w3 = __int_as_float(__builtin_amdgcn_readlane(ew, j + 3));
        const int b0 = s0 >> 8, b1 = s1 >> 8, b2 = s2 >> 8, b3 = s3 >> 8;
        const bool distinct = b0 != b1 && b0 != b2 && b0 != b3 && b1 != b2 && b1 != b3 && b2 != b3;   // scalar unit
        float* a0 = reinterpret_cast<float*>(acc_lane + __umul24((unsigned)b0, col_mul4));
        float* a1 = reinterpret_cast<float*>(acc_lane + __umul24((unsigned)b1, col_mul4));
        float* a2 = reinterpret_cast<float*>(acc_lane + __umul24((unsigned)b2, col_mul4));
        float* a3 = reinterpret_cast<float*>(acc_lane + __umul24((unsigned)b3, col_mul4));
        if (distinct) {
          const float v0 = *a0, v1 = *a1, v2 = *a2, v3 = *a3;
          *a0 = v0 + g0 * w0; *a1 = v1 + g1 * w1; *a2 = v2 + g2 * w2; *a3 = v3 + g3 * w3;
        } else {
          *a0 += g0 * w0; *a1 += g1 * w1; *a2 += g2 * w2; *a3 += g3 * w3;
        }
      }
#pragma unroll 1
      for (; j < cnt; ++j) {
        const int s0 = __builtin_amdgcn_readlane(e, j);
        add1(s0, __int_as_float(__builtin_amdgcn_readlane(ew, j)), gv_lane[(s0 & 0xff) * gvw]);
      }
    }
  };

  // staged chunk -> this lane's column edges (registers); restage: the chunk is not the one the prologue staged
  auto enter_chunk = [&](int c0, int n, bool restage) {
    if (restage) {
      __syncthreads();
      for (int i = tid; i < cap; i += NW * 64) eimg[i] = 0;
      __syncthreads();
      stage(c0, n);
      __syncthreads();
    }
    // A RoI whose gradient rows are all zeros adds nothing anywhere (background RoIs in the box head; the zero-padded ground-truth
    // rows of the tracking RoIs -- (0,0,0,0) boxes whose 49 bins ALL cover pixel (0, 0)): it belongs to no image
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {
      const int rl = min(rd * 64 + lane, cap - 1);
      const bool on = rd * 64 + lane < n && eimg[rl] != 0 && eown[rl] != 0;
#pragma unroll
      for (int k = 0; k < P; ++k) cw[rd][k] = on ? ecol[k * cap + rl] : 0;   // (start = end = 0: contains no pixel)
    }
    cur_h = -1;
    PB_STAMP(4);
  };
  // The pixel loop, instantiated twice: `single` = the image's RoIs are ONE staged chunk (staged before the loop: no staging code, no
  // barrier and none of its registers inside the loop -- the common case); otherwise every pixel walks the chunks, restaging each.
  auto pixels = [&](auto single) {
  constexpr bool kTickets = decltype(single)::value;
  // One staged chunk: the workgroup's NW * ppw pixels (g + G * k, k = 0 .. NW * ppw - 1) are taken by TICKET -- a wave that owns a
  // pixel under a stack of RoIs (thousands of list entries, walked by that one wave) leaves the others to the rest of the workgroup:
  // with a fixed three pixels per wave such workgroups ran 41 - 44 us against a median of 27 with the same number of entries.  With
  // several chunks every wave must pass the same barriers: pixel k = wave + NW * it.
  int h = mdiv(p_first, a.width, a.width_magic), w = p_first - h * a.width;          // (no division per pixel)
  int ticket = wave;
  for (int it = 0; kTickets ? ticket < NW * a.ppw : it < a.ppw; ++it) {
    int p = p_first + it * p_step;
    if constexpr (kTickets) {
      p = p_first - a.wgs_per_image * wave + a.wgs_per_image * ticket;
      h = mdiv(p, a.width, a.width_magic); w = p - h * a.width;
      int next = 0;
      if (lane == 0) next = atomicAdd(&run[2], 1);
      ticket = __builtin_amdgcn_readfirstlane(next);         // (the NEXT pixel's ticket, on its way while this pixel is worked on)
    } else {
      if (it > 0) { h += step_h; w += step_w; if (w >= a.width) { w -= a.width; ++h; } }
    }
    const bool valid = p < hw;
    int nlist = 0;
    // the compact gradient of the pixel is requested now and added in front of the write-out (it sat on the critical path of every
    // pixel as a dependent global load: 1.8 k of a pixel's 6 k cycles)
    float addv[4] = {0.f, 0.f, 0.f, 0.f};
    const bool add_pre = a.add && a.add_count <= 256 && valid;
    if (add_pre) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < a.add_count) addv[u] = a.add[((long)b * hw + p) * a.add_count + lane + 64 * u];
    }
    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = r_first + ch * cap;
      const int n = min(cap, r_end - c0);
      if constexpr (!decltype(single)::value) enter_chunk(c0, n, true);
      if (!valid || (a.ablate & 8)) continue;
      if (h != cur_h) {
        cur_h = h;
#pragma unroll
        for (int rd = 0; rd < NR; ++rd) {
          const int rl = min(rd * 64 + lane, cap - 1);
          int lo = 0, hi1 = 0;
          if (rd * 64 < n) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
              const int e = erow[k * cap + rl];
              hi1 += (e & 0xffff) <= h ? 1 : 0;     // starts <= h: a prefix of the bins
              lo += (e >> 16) <= h ? 1 : 0;         // ends <= h: the bins before the first one that still contains h
            }
          }
          prow[rd] = lo | (hi1 << 4);
        }
      }
      // ---- phase 1: lane's RoI rd*64 + lane -> the rectangle of its bins that contains (h, w): plo | phi1 << 4 | qlo << 8 | qhi1 << 12
      int rect[NR];
#pragma unroll
      for (int rd = 0; rd < NR; ++rd) {
        int qlo = 0, qhi1 = 0;
        if (rd * 64 < n) {
#pragma unroll
          for (int k = 0; k < P; ++k) {
            qhi1 += (cw[rd][k] & 0xffff) <= w ? 1 : 0;
            qlo += (cw[rd][k] >> 16) <= w ? 1 : 0;
          }
        }
        rect[rd] = prow[rd] | (qlo << 8) | (qhi1 << 12);
      }
      PB_STAMP(5 + 6 * it);
#pragma unroll 1
      for (int rd = 0; rd * 64 < n; ++rd) {
        int rc = rect[0];
#pragma unroll
        for (int q = 1; q < NR; ++q) rc = rd == q ? rect[q] : rc;
        const int plo = rc & 15, phi1 = (rc >> 4) & 15, qlo = (rc >> 8) & 15, qhi1 = rc >> 12;
        const int nh = (phi1 > plo && qhi1 > qlo) ? (phi1 - plo) * (qhi1 - qlo) : 0;   // (masked RoIs: zero column edges, nh = 0)
        const int incl = wave_incl_scan(nh);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0 || (a.ablate & 1)) continue;
        // the round's entries go to the list in lane (= RoI) order, as many whole lanes as fit; a full list is walked and refilled
        // (a pixel under a stack of small RoIs -- all 49 bins of each on it -- lists thousands of entries; walking them RoI by RoI,
        // every lane on the same rectangle, made the workgroups that owned such pixels the launch's stragglers: 69 against 26 us)
        const int rl = rd * 64 + lane;
        auto weight = [&](unsigned area) { return area < (unsigned)kWTab ? wtab[area] : inv_bins / (float)area; };
        int done = 0;
#pragma unroll 1
        while (done < total) {
          const bool take = nh > 0 && incl - nh >= done && incl - done <= kListCap - nlist;
          const unsigned long long tm = __ballot(take);
          if (tm == 0ULL) { walk(nlist); nlist = 0; continue; }     // (a lane lists at most P * P = 49 <= kListCap entries: it fits an empty list)
          const int upto = __builtin_amdgcn_readlane(incl, 63 - (int)__builtin_clzll(tm));
          if (take) {
            int pos = nlist + incl - nh - done;
            for (int ph = plo; ph < phi1; ++ph) {
              const int er = erow[ph * cap + rl];
              const unsigned hgt = (unsigned)((er >> 16) - (er & 0xffff));
              for (int pw = qlo; pw < qhi1; ++pw) {
                const int ec = ecol[pw * cap + rl];
                list_rb[pos] = rl | ((ph * P + pw) << 8);
                list_w[pos] = weight(__umul24(hgt, (unsigned)((ec >> 16) - (ec & 0xffff))));
                ++pos;
              }
            }
          }
          nlist += upto - done;
          done = upto;
        }
      }
      PB_STAMP(6 + 6 * it);
      if (a.ablate & 2) nlist = 0;
#ifdef DTT_PSROI_BWD_STAMP
      if (blockIdx.x == 0 && tid == 0 && it < 4) dtt_psroi_bwd_stamps[56 + it] = (unsigned long long)nlist;   // (entries walked)
#endif
      walk(nlist);
      nlist = 0;
      PB_STAMP(7 + 6 * it);
    }
    if (!valid || (a.ablate & 4)) continue;
    // ---- write-out: the whole row of the pixel, the accumulator zeroed behind the reads
    const long px = (long)b * hw + p;
    if (add_pre) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < a.add_count) acc[a.add_first + lane + 64 * u] += addv[u];
    } else if (a.add) {
      for (int c = lane; c < a.add_count; c += 64) acc[a.add_first + c] += a.add[px * a.add_count + c];
    }
    PB_STAMP(8 + 6 * it);
    float* dst = a.gmap + px * a.pixel_stride;
    if (((a.pixel_stride | a.row_floats) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gmap) & 15) == 0) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      const int n4 = a.row_floats / 4;
#pragma unroll 1
      for (int j0 = lane; j0 < n4; j0 += 64 * kWOut) {      // kWOut 16-byte pieces per lane in flight
        f32x4 v[kWOut];
#pragma unroll
        for (int u = 0; u < kWOut; ++u) v[u] = reinterpret_cast<f32x4*>(acc)[min(j0 + 64 * u, n4 - 1)];
#pragma unroll
        for (int u = 0; u < kWOut; ++u)
          if (j0 + 64 * u < n4) {
            __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(dst) + j0 + 64 * u);
            reinterpret_cast<f32x4*>(acc)[j0 + 64 * u] = zero;
          }
      }
    } else {
      for (int j = lane; j < a.row_floats; j += 64) { dst[j] = acc[j]; acc[j] = 0.f; }
    }
    PB_STAMP(9 + 6 * it);
  }
  };
#ifdef DTT_PSROI_BWD_STAMP
  if (tid == 0 && blockIdx.x < 1024) dtt_psroi_bwd_cnt[blockIdx.x * 4 + 3] = (unsigned)nchunks | (guessed ? 0x100u : 0u) | ((unsigned)(r_end - r_first) << 16);
#endif
  if (nchunks <= 1) {
    if (nchunks == 1) enter_chunk(r_first, min(cap, r_end - r_first), !guessed);
    pixels(std::true_type{});
  } else {
    pixels(std::false_type{});
  }
  PB_STAMP(50);
#ifdef DTT_PSROI_BWD_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 1024) dtt_psroi_bwd_wg[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) dtt_psroi_bwd_stamps[(blockIdx.x == 0 ? 0 : 2) * 64 + 63] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int P, int NW, int NR>
int launch_rows(const PmBwd& a, int batch_size, size_t lds, hipStream_t stream) {
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(psroi_pm_bwd_rows_kernel<P, NW, NR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    DTT_REQUIRE(e == hipSuccess, "psroi_pm backward: cannot raise dynamic LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL((psroi_pm_bwd_rows_kernel<P, NW, NR>), dim3(batch_size * a.wgs_per_image), dim3(NW * 64), lds, stream, a);
  return 1;
}

}  // namespace

#ifdef DTT_PSROI_BWD_STAMP
extern "C" int dtt_psroi_bwd_stamps_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_psroi_bwd_stamps), sizeof(unsigned long long) * n) == hipSuccess;
}
extern "C" int dtt_psroi_bwd_cnt_read(unsigned int* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_psroi_bwd_cnt), sizeof(unsigned int) * n) == hipSuccess;
}
extern "C" int dtt_psroi_bwd_wg_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_psroi_bwd_wg), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

__attribute__((visibility("hidden"))) int dtt_psroi_pm_backward_old(const float* grad_vote, const float* rois, int num_rois, int batch_size,
                                                                    int height, int width, int pooled, float spatial_scale, int output_dim,
                                                                    int cp, long pixel_stride, float* grad_map, int* edges, hipStream_t stream);

// Backward of the votes of up to two heads pooled from one position-major map (dtt_psroi_pm_forward / dtt_psroi_pm_det_forward):
// head h has output_dim_h columns padded to cp_h per bin; head 0's bins start at column 0 of a pixel's row, head 1's at
// pooled^2 * cp0.  Writes columns [0, row_floats) of EVERY pixel (zeros where no RoI reaches and in the padding columns past
// the heads): no pre-zeroing, no scratch, ONE launch.  add_cols (pixels, add_count) or NULL: a compact gradient added into columns
// [add_first, add_first + add_count) after the pooling sums (the second consumer of the map under autograd).
extern "C" int dtt_psroi_pm_backward_heads(const float* grad_vote0, int output_dim0, int cp0, const float* grad_vote1, int output_dim1,
                                           int cp1, const float* rois, int num_rois, int batch_size, int height, int width, int pooled,
                                           float spatial_scale, long pixel_stride, int row_floats, const float* add_cols, int add_first,
                                           int add_count, float* grad_map, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(batch_size > 0 && height > 0 && width > 0 && pooled > 0 && output_dim0 > 0 && num_rois >= 0, "psroi_pm backward: bad shape");
  DTT_REQUIRE(height < 32768 && width < 32768, "psroi_pm backward: map side above 32767");
  if (!grad_vote1) { output_dim1 = 0; cp1 = 0; }
  DTT_REQUIRE(cp0 >= output_dim0 && cp1 >= output_dim1 && cp0 + cp1 <= 64, "psroi_pm backward: %d + %d columns per bin (at most 64 together)", cp0, cp1);
  DTT_REQUIRE((long)pooled * pooled * (cp0 + cp1) <= row_floats && row_floats <= pixel_stride,
              "psroi_pm backward: %d bins x %d do not fit the row of %d (stride %ld)", pooled * pooled, cp0 + cp1, row_floats, pixel_stride);
  DTT_REQUIRE(grad_map && (num_rois == 0 || (grad_vote0 && rois)), "psroi_pm backward: null pointer");
  DTT_REQUIRE(!add_cols || (add_first >= 0 && add_count > 0 && add_first + add_count <= row_floats), "psroi_pm backward: added columns outside the row");
  DTT_REQUIRE(pooled == 7, "psroi_pm backward: pooled size %d not instantiated (7)", pooled);
  PmBwd a;
  a.gv0 = grad_vote0; a.gv1 = grad_vote1; a.od0 = output_dim0; a.cp0 = cp0; a.od1 = output_dim1; a.cp1 = cp1;
  a.rois = rois; a.num_rois = num_rois; a.spatial_scale = spatial_scale; a.batch_size = batch_size;
  a.height = height; a.width = width; a.pixel_stride = pixel_stride;
  a.row_floats = row_floats; a.add = add_cols; a.add_first = add_first; a.add_count = add_count; a.gmap = grad_map;
  // RoIs staged per chunk: the per-image share when callers list the same number per image (training: 128), more chunks otherwise
  const int per_image = dtt_cdiv(num_rois > 0 ? num_rois : 1, batch_size);
  a.cap = std::min(64 * kMaxRounds, dtt_cdiv(per_image, 64) * 64);
  a.per_image = per_image;
  static const int env_nw = getenv("DTT_PSROI_BWD_WAVES") ? atoi(getenv("DTT_PSROI_BWD_WAVES")) : 0;   // developer sweeps
  const int acc_floats = (row_floats + 3) & ~3;
  const size_t shared = ((size_t)a.cap * (cp0 + cp1) + (size_t)(2 * pooled + 2) * a.cap + kWTab + 4) * 4;
  auto lds_of = [&](int nw) { return (size_t)nw * (acc_floats + 2 * kListCap + 4) * 4 + shared; };
  int nw = env_nw ? env_nw : 16;
  DTT_REQUIRE(nw == 4 || nw == 8 || nw == 16, "psroi_pm backward: DTT_PSROI_BWD_WAVES must be 4, 8 or 16");
  if (a.cap > 128 && nw == 16) nw = 8;   // (four rounds of RoIs in registers: more than what 16 waves leave each lane)
  while (nw > 4 && lds_of(nw) > 160 * 1024) nw >>= 1;
  DTT_REQUIRE(lds_of(nw) <= 160 * 1024, "psroi_pm backward: row of %d floats does not fit LDS", row_floats);
  // pixels per wave: one round of workgroups over the CUs (what fits a CU at once), so the prologue is paid once
  const int hw = height * width;
  const int cus = dtt_device_cus();
  const int wg_per_cu = (int)std::max((size_t)1, std::min((size_t)(2048 / (nw * 64)), (size_t)(160 * 1024) / lds_of(nw)));
  static const int env_ppw = getenv("DTT_PSROI_BWD_PPW") ? atoi(getenv("DTT_PSROI_BWD_PPW")) : 0;
  const long slots = (long)cus * wg_per_cu * nw;
  // (at least two pixels per wave: every workgroup pays the same prologue, and a launch whose pixels are cheap -- the tracking head's
  //  handful of RoIs -- is all prologue: 16.2 us as 1274 four-wave workgroups of one pixel per wave, 12.5 us as 160 of these)
  a.ppw = env_ppw > 0 ? env_ppw : std::max(2, dtt_cdiv((long)batch_size * hw, slots));
  a.wgs_per_image = dtt_cdiv(hw, nw * a.ppw);
  a.ppw = dtt_cdiv(hw, nw * a.wgs_per_image);       // (round robin: the rounds that reach a pixel)
  static const int env_ablate = getenv("DTT_PSROI_BWD_ABLATE") ? atoi(getenv("DTT_PSROI_BWD_ABLATE")) : 0;
  a.ablate = env_ablate;
  const long n_max = std::max((long)hw + (long)nw * a.wgs_per_image, (long)batch_size * a.wgs_per_image);
  a.width_magic = n_max * width < (1L << 32) && width > 1 ? (unsigned)((1ULL << 32) / (unsigned)width + 1) : 0u;
  a.wpi_magic = n_max * a.wgs_per_image < (1L << 32) && a.wgs_per_image > 1 ? (unsigned)((1ULL << 32) / (unsigned)a.wgs_per_image + 1) : 0u;
  const size_t lds_total = lds_of(nw);
  dtt_prof_begin("psroi_pm_bwd", stream);   // (event tag: the one launch)
  int ok;
  const int nr = a.cap / 64;   // 1, 2, 3 (run as 4) or 4 rounds of 64 RoIs per chunk
#define DTT_PMB_ROWS(NWV)                                                                              \
  (nr == 1 ? launch_rows<7, NWV, 1>(a, batch_size, lds_total, stream) : launch_rows<7, NWV, 2>(a, batch_size, lds_total, stream))
  if (nr > 2) ok = nw == 8 ? launch_rows<7, 8, 4>(a, batch_size, lds_total, stream) : launch_rows<7, 4, 4>(a, batch_size, lds_total, stream);
  else if (nw == 16) ok = DTT_PMB_ROWS(16);
  else if (nw == 8) ok = DTT_PMB_ROWS(8);
  else ok = DTT_PMB_ROWS(4);
#undef DTT_PMB_ROWS
  if (!ok) return 0;
  dtt_prof_end("psroi_pm_bwd", stream);
  DTT_CHECK_LAUNCH("psroi_pm_bwd");
  return 1;
}

// One head (the ABI of rounds 4 - 5): columns [0, pooled^2 * cp) of every pixel.
extern "C" int dtt_psroi_pm_backward(const float* grad_vote, const float* rois, int num_rois, int batch_size, int height, int width,
                                     int pooled, float spatial_scale, int output_dim, int cp, long pixel_stride, float* grad_map,
                                     int* edges, void* stream_) {
  const char* env_old = getenv("DTT_PSROI_BWD_OLD");   // (read per call: the A/B test flips it inside one process)
  const bool old = env_old && atoi(env_old) != 0;
  if (old || pooled != 7)
    return dtt_psroi_pm_backward_old(grad_vote, rois, num_rois, batch_size, height, width, pooled, spatial_scale, output_dim, cp, pixel_stride,
                                     grad_map, edges, static_cast<hipStream_t>(stream_));
  return dtt_psroi_pm_backward_heads(grad_vote, output_dim, cp, nullptr, 0, 0, rois, num_rois, batch_size, height, width, pooled, spatial_scale,
                                     pixel_stride, pooled * pooled * cp, nullptr, 0, 0, grad_map, stream_);
}

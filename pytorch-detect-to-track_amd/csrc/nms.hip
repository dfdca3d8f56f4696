// NMS for gfx950: 64x64 IoU bitmask tiles (four wave64s per tile, one uint64 per row) + an on-device
// greedy sweep, so nothing returns to the host.
//
// Replaces nms_kernel + nms_cuda_compute (reference nms/src/nms_cuda_kernel.cu:41-161).  The IoU
// arithmetic (devIoU, .cu:31-39) is reproduced operation by operation with FP contraction off, so the
// keep list is bit-exact with the oracle (oracle/dtt_oracle.c: oracle_nms).
#include "common.h"
#include "nms_internal.h"

namespace {

constexpr int kTile = 64;          // threadsPerBlock = sizeof(unsigned long long) * 8 (.cu:29)
constexpr int kSuper = 1024;       // boxes per LDS-resident super-chunk of the sweep
constexpr int kSuperWords = kSuper / kTile;  // 16
constexpr int kRowStride = kSuperWords + 1;  // 17 words: conflict-free ds_read_b64 down a column
constexpr int kSweepThreads = 1024;

// devIoU(a, b) > thresh (.cu:31-39, 101: same operations, same order, no contraction -- the file is built with
// -ffp-contract=off) with the division skipped when no lane of the wave needs it.  fl(inter / uni) > thresh is decided by
// inter against thresh * uni whenever the two are further apart than the roundings involved (1e-6 relative against 2^-23 for the
// product and the quotient together): the common case -- disjoint boxes have inter == 0 -- costs two multiplies and two compares
// instead of a correctly rounded division; a wave in which any lane is within that margin (or has a non-positive / non-finite
// union) takes the reference's division for all its lanes.  The same answer as devIoU(a, b) > thresh in every case.
// Sb = (b2 - b0 + 1) * (b3 - b1 + 1), the column box's area, is made once per staged column (the same expression, the same bits).
__device__ __forceinline__ bool iou_over(const float a0, const float a1, const float a2, const float a3, const float Sa,
                                         const float b0, const float b1, const float b2, const float b3, const float Sb, const float thresh) {
  // max / min of two coordinates as the bare instructions: fmaxf / fminf make the compiler put a canonicalising v_max_f32 x, x in front
  // for every value that comes out of LDS (4 of the 33 vector instructions per IoU test; the mask kernel is VALU-bound, and it even
  // folds v_med3_f32 with an infinity back into that form).  For non-NaN inputs v_max_f32 / v_min_f32 return one of their operands.
  float left, right, top, bottom;
  asm("v_max_f32 %0, %1, %2" : "=v"(left) : "v"(a0), "v"(b0));
  asm("v_min_f32 %0, %1, %2" : "=v"(right) : "v"(a2), "v"(b2));
  asm("v_max_f32 %0, %1, %2" : "=v"(top) : "v"(a1), "v"(b1));
  asm("v_min_f32 %0, %1, %2" : "=v"(bottom) : "v"(a3), "v"(b3));
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  const float uni = Sa + Sb - interS;
  const float p = thresh * uni;
  const bool sure_hit = interS > p * 1.000001f, sure_miss = interS < p * 0.999999f;
  const bool decided = thresh > 0.f && uni > 1e-30f && uni < 1e30f && (sure_hit || sure_miss);
  if (__all(decided)) return sure_hit;
  return interS / uni > thresh;
}

constexpr int kMaskWaves = 4;      // waves per 64 x 64 tile: each takes 16 of the tile's 64 columns
constexpr int kMaskThreads = kMaskWaves * kTile;
constexpr int kStateWords = 8;     // sweep state parked behind the bit matrix: [0] done, [1] survivors, [2 ..) removal words

// Words per image behind the n_max x col_blocks bit matrix: the parked sweep state, then one "lower" word per box --
// the bits of the box's own 64-box chunk BELOW its own position (bit i of lower[r]: box chunk(r) * 64 + i, i < r % 64,
// overlaps box r; devIoU is symmetric bit for bit, so this is column r of the diagonal tile).  The sweep settles a chunk
// from these with a few wave-wide steps instead of one scalar step per box.
__host__ __device__ inline long mask_state_offset(int n_max, int col_blocks) { return (long)n_max * col_blocks; }
__host__ __device__ inline long mask_lower_offset(int n_max, int col_blocks) { return (long)n_max * col_blocks + col_blocks + kStateWords; }

// grid (col_blocks, row_blocks, batch), block 256 = 4 waves: wave q of a tile computes columns 16q .. 16q+15 of every row
// and stores its 16 bits of the row's word (the serial column loop is what bounds a tile: 64 dependent IoUs with a true
// division each).  Only tiles with col >= row are produced: the sweep never reads words left of a row's own block
// (nms_cuda_kernel.cu:139 starts at j = nblock).
// A workgroup owns column block col_block0 + blockIdx.x and the row blocks row_block0 + blockIdx.y, + gridDim.y, ... < row_block_end
// (one tile per workgroup in the usual launch; the second phase of a two-phase NMS uses a short grid that loops, so that
// the images the first phase finished cost a few hundred workgroup exits instead of tens of thousands).
__global__ __launch_bounds__(kMaskThreads) void nms_mask_kernel(const float* __restrict__ boxes, int boxes_dim,
                                                                long box_batch_stride, const int* __restrict__ n_per_image,
                                                                int n_max, float thresh, unsigned long long* __restrict__ mask,
                                                                long mask_batch_stride, int col_blocks, int row_block0,
                                                                int row_block_end, int col_block0, int check_done,
                                                                const int* __restrict__ keep, long keep_batch_stride,
                                                                int kept_row_limit) {
  const int col_start = blockIdx.x + col_block0, img = blockIdx.z;
  // second phase of a two-phase NMS: nothing to do for an image whose sweep already has max_keep survivors
  if (check_done && mask[img * mask_batch_stride + mask_state_offset(n_max, col_blocks)] != 0ULL) return;
  const int n_boxes = n_per_image ? n_per_image[img] : n_max;
  if (col_start * kTile >= n_boxes) return;
  const float* b = boxes + img * box_batch_stride;
  unsigned long long* m = mask + img * mask_batch_stride;
  unsigned long long* lower = m + mask_lower_offset(n_max, col_blocks);
  const int col_size = min(n_boxes - col_start * kTile, kTile);
  __shared__ float bb[kTile * 4];
  __shared__ float bs[kTile];      // areas of the staged column boxes
  const int t = threadIdx.x & (kTile - 1), q = threadIdx.x >> 6;
  const int i0 = q * (kTile / kMaskWaves), i1 = min(i0 + kTile / kMaskWaves, col_size);
  bool staged = false;
  auto stage_columns = [&]() {   // (uniform: the callers' loop bounds do not depend on the thread)
    if (staged) return;
    if (threadIdx.x < col_size) {
      const float* p = b + (long)(kTile * col_start + threadIdx.x) * boxes_dim;
      bb[threadIdx.x * 4 + 0] = p[0];
      bb[threadIdx.x * 4 + 1] = p[1];
      bb[threadIdx.x * 4 + 2] = p[2];
      bb[threadIdx.x * 4 + 3] = p[3];
      bs[threadIdx.x] = (p[2] - p[0] + 1) * (p[3] - p[1] + 1);
    }
    __syncthreads();
    staged = true;
  };
  if (keep) {
    // Second phase, rows of the FIRST phase's boxes (below kept_row_limit): the sweep only ever reads the rows of boxes it kept
    // (the replay over the new columns), so only those are computed -- in groups of 64 entries of the image's keep list -- and the
    // rows of suppressed boxes stay unwritten.  The row blocks from kept_row_limit / 64 on follow below as usual.
    const int kept = (int)m[mask_state_offset(n_max, col_blocks) + 1];
    const int* kl = keep + img * keep_batch_stride;
    for (int g0 = blockIdx.y * kTile; g0 < kept; g0 += gridDim.y * kTile) {
      stage_columns();
      if (g0 + t < kept) {
        const int cur = kl[g0 + t];
        const float* p = b + (long)cur * boxes_dim;
        const float a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
        const float Sa = (a2 - a0 + 1) * (a3 - a1 + 1);
        unsigned bits = 0;
        for (int i = i0; i < i1; ++i) {
          if (iou_over(a0, a1, a2, a3, Sa, bb[i * 4 + 0], bb[i * 4 + 1], bb[i * 4 + 2], bb[i * 4 + 3], bs[i], thresh))
            bits |= 1u << (i - i0);
        }
        reinterpret_cast<unsigned short*>(m + (long)cur * col_blocks + col_start)[q] = (unsigned short)bits;
      }
    }
    row_block0 = max(row_block0, kept_row_limit / kTile);
  }
  for (int row_start = row_block0 + blockIdx.y; row_start < row_block_end && row_start <= col_start; row_start += gridDim.y) {
    if (row_start * kTile >= n_boxes) break;
    stage_columns();
    const int row_size = min(n_boxes - row_start * kTile, kTile);
    if (t < row_size) {
      const int cur = kTile * row_start + t;
      const float* p = b + (long)cur * boxes_dim;
      const float a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
      const float Sa = (a2 - a0 + 1) * (a3 - a1 + 1);
      unsigned bits = 0;
      for (int i = i0; i < i1; ++i) {
        if (iou_over(a0, a1, a2, a3, Sa, bb[i * 4 + 0], bb[i * 4 + 1], bb[i * 4 + 2], bb[i * 4 + 3], bs[i], thresh))
          bits |= 1u << (i - i0);
      }
      unsigned short* word = reinterpret_cast<unsigned short*>(m + (long)cur * col_blocks + col_start);
      if (row_start == col_start) {
        // diagonal tile: the row's word keeps the bits above its own position (.cu:98 starts at threadIdx.x + 1), the bits
        // below go to the box's "lower" word
        const unsigned long long full = (unsigned long long)bits << i0;
        const unsigned long long above = (t == kTile - 1) ? 0ULL : (~0ULL << (t + 1));
        const unsigned long long below = (1ULL << t) - 1ULL;
        word[q] = (unsigned short)((full & above) >> i0);
        reinterpret_cast<unsigned short*>(lower + cur)[q] = (unsigned short)((full & below) >> i0);
      } else {
        word[q] = (unsigned short)bits;
      }
    }
  }
}

#ifdef DTT_NMS_TRACE   // developer build (tools/nms_phases.py): shader cycles of the sweep's phases, summed per image and launch
__device__ unsigned long long dtt_nms_cycles[64 * 8];
#define NMS_T0() unsigned long long t_prev = __builtin_readcyclecounter()
#define NMS_ACC(slot) do { const unsigned long long t_now = __builtin_readcyclecounter(); \
    if (threadIdx.x == 0 && blockIdx.x < 32) dtt_nms_cycles[((sc_begin > 0 ? 32 : 0) + blockIdx.x) * 8 + (slot)] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define NMS_T0() do {} while (0)
#define NMS_ACC(slot) do {} while (0)
#endif
#ifdef DTT_NMS_TRACE   // inside wave 0's walk: slots 4 (settling a chunk), 5 (keep-list entries), 6 (row OR + quarter reduction), 7 (fixpoint iterations)
#define NMS_B0() unsigned long long tb_prev = __builtin_readcyclecounter()
#define NMS_BACC(slot) do { const unsigned long long tb_now = __builtin_readcyclecounter(); \
    if (lane == 0 && blockIdx.x < 32) dtt_nms_cycles[((sc_begin > 0 ? 32 : 0) + blockIdx.x) * 8 + (slot)] += tb_now - tb_prev; tb_prev = tb_now; } while (0)
#define NMS_BCOUNT(slot, v) do { if (lane == 0 && blockIdx.x < 32) dtt_nms_cycles[((sc_begin > 0 ? 32 : 0) + blockIdx.x) * 8 + (slot)] += (v); } while (0)
#else
#define NMS_B0() do {} while (0)
#define NMS_BACC(slot) do {} while (0)
#define NMS_BCOUNT(slot, v) do {} while (0)
#endif

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Greedy sweep (nms_cuda_kernel.cu:131-144) for one image per workgroup.
//   for i ascending: if bit i of remv is clear -> keep i, remv |= mask[i][i/64 ...]
// restructured so that global-memory latency is off the serial path (per 1024-box super-chunk; a software pipeline, see the loop):
//   A. the 1024 x 16-word diagonal super-block of the mask and the boxes' "lower" words, loaded into registers while the
//      PREVIOUS super-chunk was walked, go to LDS;
//   B. wave 0 walks the 16 chunks of 64 boxes: a chunk is settled from its lower words by a wave-wide fixpoint (a handful of
//      ballots), then lane (word, quarter) ORs that word of the chunk's kept rows -- all 16 rows of the quarter read from LDS
//      unconditionally and masked -- and the quarters meet through v_permlane swaps; nothing in the walk touches global memory.
//      Beside it waves 1 .. 15 OR the previous super-chunk's kept rows over the columns beyond the next super-chunk (C_far);
//   C. everybody ORs the kept rows over the 16 words of the next super-chunk (C_near) and writes the keep-list entries.
// Optionally writes the surviving boxes straight into the RoI tensor (proposal layer epilogue).
//
// Two-phase use (sc_begin / sc_end = range of 1024-box super-chunks; dtt_nms_batched_launch): when only max_keep << n
// survivors are wanted, the mask rows of the first few super-chunks are computed and swept first; the sweep state (removal
// words, survivor count) is parked behind the image's mask matrix, state[0] tells the second mask launch and the second
// sweep whether there is anything left to do.  The keep list itself lives in keep_out all along.
__global__ __launch_bounds__(kSweepThreads) void nms_sweep_kernel(
    unsigned long long* __restrict__ mask, long mask_batch_stride, const int* __restrict__ n_per_image,
    int n_max, int col_blocks, int max_keep, int* __restrict__ keep_out, long keep_batch_stride,
    int* __restrict__ num_out, const float* __restrict__ boxes, int boxes_dim, long box_batch_stride,
    float* __restrict__ rois_out, int rois_rows, int sc_begin, int sc_end, int two_phase) {
  // (two_phase, first phase: only mask columns < sc_end * 16 exist yet -- the square the sweep of these super-chunks reads;
  //  second phase: the rows kept so far are replayed into the removal words of the columns that were computed since)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* sb = reinterpret_cast<unsigned long long*>(smem);            // [kSuper][kRowStride]
  unsigned long long* remv = sb + (size_t)kSuper * kRowStride;                      // [col_blocks]
  unsigned long long* lowbuf = remv + col_blocks;                                   // [kSuper]: the "lower" words of the staged super-chunk
  int* kept_list = reinterpret_cast<int*>(lowbuf + kSuper);                         // [2][kSuper]: this super-chunk's and the previous one's
  int* ctl = kept_list + 2 * kSuper;                                                // [0]=nk, [1]=total, [2]=done

  const int img = blockIdx.x;
  const int n = n_per_image ? n_per_image[img] : n_max;
  const unsigned long long* m = mask + img * mask_batch_stride;
  unsigned long long* state = mask + img * mask_batch_stride + mask_state_offset(n_max, col_blocks);   // [0] done, [1] total, [2..] remv
  const unsigned long long* lower = mask + img * mask_batch_stride + mask_lower_offset(n_max, col_blocks);
  int* keep = keep_out ? keep_out + img * keep_batch_stride : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = (n + kTile - 1) / kTile;
  const int limit = (max_keep > 0 && max_keep < n) ? max_keep : n;
  NMS_T0();

  if (sc_begin > 0) {
    if (state[0] != 0ULL) return;   // the first phase finished this image (and wrote its outputs)
    const int kept_so_far = (int)state[1];
    const int w_first = sc_begin * kSuperWords;   // first column block the first phase did not have
    for (int j = tid; j < col_blocks; j += kSweepThreads) remv[j] = state[2 + j];
    if (tid == 0) { ctl[0] = 0; ctl[1] = kept_so_far; ctl[2] = 0; }
    // replay: every row kept in phase 1 suppresses across the columns computed since.  The kept rows are dealt to as many
    // thread slices per column as the workgroup has threads for (a column alone would walk up to max_keep rows serially:
    // ADVICE r2) and the slices meet with an LDS atomic OR -- order-free, so still deterministic.
    const int ncol = cb - w_first;
    if (ncol > 0 && kept_so_far > 0) {
      __syncthreads();
      const int nsl = max(1, kSweepThreads / ncol);
      const int c = tid % ncol, sl = tid / ncol;
      if (sl < nsl) {
        const int j = w_first + c;
        unsigned long long r = 0ULL;
        for (int k = sl; k < kept_so_far; k += nsl) r |= m[(long)keep[k] * col_blocks + j];
        if (r) atomicOr(&remv[j], r);
      }
    }
  } else {
    for (int j = tid; j < col_blocks; j += kSweepThreads) remv[j] = 0;
    if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; }
  }
  __syncthreads();
  NMS_ACC(0);   // set-up / replay of the first phase's kept rows

  const int n_super = (n + kSuper - 1) / kSuper;
  const int sc_last = min(sc_end, n_super);
  const int col_end = (two_phase && sc_begin == 0) ? min(cb, sc_end * kSuperWords) : cb;   // columns that exist in this phase
  // Software pipeline over the super-chunks (round 5; the serial walk B of wave 0 is what cannot be shortened):
  //   * the diagonal super-block of super-chunk sc + 1 is LOADED into registers (16 words per thread, one round trip) while wave 0
  //     walks super-chunk sc, and only written to LDS once sc is done with the staging area -- the staging latency (a quarter of the
  //     sweep: profiles/r05_nms_sweep_phases_before.txt) disappears under B;
  //   * the rows kept in super-chunk sc suppress boxes of ALL later super-chunks, but only the 16 words of super-chunk sc + 1 are
  //     needed before its walk starts: those (C_near) are OR-ed by the whole workgroup right after B, the other columns (C_far) by
  //     waves 1 .. 15 WHILE wave 0 walks super-chunk sc + 1 (its kept list lives in the other half of a double buffer; the words
  //     C_far touches lie beyond the ones that walk reads).
  // The recursion is unchanged -- the same rows are OR-ed into the same words before anybody reads them -- so the keep list is too.
  constexpr int NL = 16;   // words per thread: a full 1024 x 16-word super-block is ONE round of loads
  unsigned long long pre[NL], pre_low;
  auto prefetch = [&](int sc) {       // diagonal super-block of super-chunk sc -> registers (unconditional, clamped loads)
    const int base = sc * kSuper, rows = min(kSuper, n - base), w0 = sc * kSuperWords, nw = min(kSuperWords, cb - w0);
    const int total = rows * nw;
    pre_low = lower[base + min(tid, rows - 1)];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int idx = min(tid + u * kSweepThreads, total - 1);
      const int r = nw == kSuperWords ? idx >> 4 : idx / nw, j = idx - r * nw;
      pre[u] = m[(long)(base + r) * col_blocks + w0 + j];
    }
  };
  auto stage = [&](int sc) {          // registers -> LDS (words left of a row's own block were never written by the mask kernel: 0)
    const int base = sc * kSuper, rows = min(kSuper, n - base), w0 = sc * kSuperWords, nw = min(kSuperWords, cb - w0);
    const int total = rows * nw;
    lowbuf[tid] = tid < rows ? pre_low : 0ULL;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int idx = tid + u * kSweepThreads;
      if (idx < total) {
        const int r = nw == kSuperWords ? idx >> 4 : idx / nw, j = idx - r * nw;
        sb[r * kRowStride + j] = ((r >> 6) <= j) ? pre[u] : 0ULL;
      }
    }
  };
  // kept rows (local numbers in kl, nk of them) of the super-chunk at row `base` OR-ed into remv[c_lo .. c_hi) by `nthr` threads,
  // this thread being number t of them: thread (column, slice) walks its share of the rows with 8 loads in flight, the slices of a
  // column meet with an LDS atomic (order-free)
  auto or_rows = [&](const int* kl, int nk, int base, int c_lo, int c_hi, int t, int nthr) {
    const int ncol = c_hi - c_lo;
    if (ncol <= 0 || nk <= 0 || t < 0) return;
    const int jt = ncol <= 16 ? 16 : ((ncol + 63) & ~63);
    const int nslice = nthr / jt;
    if (nslice < 1) {   // (more columns than threads: a strided walk, one slice)
      for (int jj = t; jj < ncol; jj += nthr) {
        unsigned long long accw = 0;
        const unsigned long long* col = m + (long)base * col_blocks + c_lo + jj;
        for (int k = 0; k < nk; ++k) accw |= col[(long)kl[k] * col_blocks];
        if (accw) atomicOr(&remv[c_lo + jj], accw);
      }
      return;
    }
    const int jj = t % jt, sl = t / jt;
    if (sl >= nslice || jj >= ncol) return;
    unsigned long long accw = 0;
    const unsigned long long* col = m + (long)base * col_blocks + c_lo + jj;
    int k = sl;
    for (; k + 7 * nslice < nk; k += 8 * nslice) {
      unsigned long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(long)kl[k + u * nslice] * col_blocks];
#pragma unroll
      for (int u = 0; u < 8; ++u) accw |= v[u];
    }
    {   // the remainder: up to seven more, all in flight together
      unsigned long long v[7];
#pragma unroll
      for (int u = 0; u < 7; ++u) v[u] = (k + u * nslice < nk) ? col[(long)kl[k + u * nslice] * col_blocks] : 0ULL;
#pragma unroll
      for (int u = 0; u < 7; ++u) accw |= v[u];
    }
    if (accw) atomicOr(&remv[c_lo + jj], accw);
  };
  int* kept_buf[2] = {kept_list, kept_list + kSuper};
  int far_nk = 0, far_base = 0, far_lo = 0, cur = 0;     // pending C_far: rows of kept_buf[cur ^ 1]
  if (sc_begin < sc_last) prefetch(sc_begin);
  for (int sc = sc_begin; sc < sc_last; ++sc) {
    const int base = sc * kSuper;
    const int rows = min(kSuper, n - base);
    const int w0 = sc * kSuperWords;
    const int nw = min(kSuperWords, cb - w0);
    int* kept_list = kept_buf[cur];
    // ---- A: the diagonal super-block, loaded during the previous walk, goes to LDS
    stage(sc);
    __syncthreads();
    NMS_ACC(1);   // A: staging
    if (sc + 1 < sc_last) prefetch(sc + 1);               // in flight under B / C_far
    // ---- C_far of the previous super-chunk (waves 1 .. 15), beside B
    if (wave != 0) or_rows(kept_buf[cur ^ 1], far_nk, far_base, far_lo, col_end, tid - 64, kSweepThreads - 64);
    // ---- B: serial part, wave 0 only.  Lanes 0..15 hold the removal words of this super-chunk in a register (R).  A chunk
    //         of 64 boxes is settled from the boxes' "lower" words L (bit i of L[j]: box i < j of the chunk overlaps box j):
    //         kept = alive & ~{ j : L[j] & kept != 0 } iterated from kept = alive.  The dependence only runs from lower to
    //         higher positions, so after k steps the first k positions are final and a repeated value is THE greedy answer
    //         (nms_cuda_kernel.cu:131-144); it takes as many steps as the longest chain of boxes that flip each other
    //         (a handful) instead of one scalar step per alive box.  Then lane (j, quarter) ORs word j of the kept rows of
    //         its quarter of the chunk (LDS reads, 4 in flight) and the quarters meet in lane j.
    if (wave == 0) {
      int total = __builtin_amdgcn_readfirstlane(ctl[1]);
      int nk = 0;
      bool done = false;
      unsigned long long R = (lane < nw) ? remv[w0 + lane] : 0ULL;
      // (nothing in the walk touches global memory: the boxes' "lower" words were staged with the block, the keep list leaves after the
      //  walk -- a store per chunk sat in front of the next chunk's load in the in-order memory counter, a full round trip per chunk)
      unsigned long long Lnext = lowbuf[lane];
      for (int c = 0; c < nw && !done; ++c) {
        const unsigned long long r = readlane64(R, c);
        const int rows_c = min(kTile, rows - c * kTile);
        const unsigned long long valid = rows_c == kTile ? ~0ULL : ((1ULL << rows_c) - 1ULL);
        const unsigned long long L = Lnext;
        if (c + 1 < nw) Lnext = lowbuf[(c + 1) * kTile + lane];   // in flight during this chunk
        const unsigned long long alive = ~r & valid;
        unsigned long long kept = alive;
        NMS_B0();
        for (;;) {
          const unsigned long long next = alive & ~__ballot((L & kept) != 0ULL);
          NMS_BCOUNT(7, 1);
          if (next == kept) break;
          kept = next;
        }
        const int room = limit - total;
        for (int extra = __builtin_popcountll(kept) - room; extra > 0; --extra)
          kept &= ~(1ULL << (63 - __builtin_clzll(kept)));
        const int nkept = __builtin_popcountll(kept);
        if ((kept >> lane) & 1ULL) {
          const int rank = __builtin_popcountll(kept & ((1ULL << lane) - 1ULL));
          kept_list[nk + rank] = c * kTile + lane;
        }
        nk += nkept;
        total += nkept;
        if (total >= limit) done = true;
        NMS_BACC(4);
        if (!done && c + 1 < nw) {
          const int j = lane & 15, quarter = lane >> 4;
          const bool owner = j > c && j < nw;
          const unsigned long long* col = sb + (size_t)(c * kTile + quarter * 16) * kRowStride + (owner ? j : 0);
          // all 16 rows of the quarter are read unconditionally (immediate offsets, two bursts of eight in flight) and masked by their
          // kept bits: no serial ctz / loop over the kept rows, one LDS latency per burst (round 5: the row-OR was two thirds of the
          // walk, profiles/r05_nms_walk_breakdown.txt)
          const unsigned kk = (unsigned)(kept >> (quarter * 16)) & 0xFFFFu;
          unsigned alo = 0, ahi = 0;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(h * 8 + u) * kRowStride];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const unsigned msk = 0u - ((kk >> (h * 8 + u)) & 1u);      // all ones for a kept row (v_bfe_i32), then one v_and_or per half
              alo |= (unsigned)v[u] & msk;
              ahi |= (unsigned)(v[u] >> 32) & msk;
            }
          }
          unsigned long long accw = ((unsigned long long)ahi << 32) | alo;
          // the four quarters (rows of 16 lanes) meet in every lane: two register swaps (v_permlane16_swap / v_permlane32_swap, VALU)
          // where __shfl_xor went through the LDS crossbar twice, back to back
          {
            unsigned lo = (unsigned)accw, hi = (unsigned)(accw >> 32);
            auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false); lo = a[0] | a[1];
            auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false); hi = b[0] | b[1];
            auto c2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false); lo = c2[0] | c2[1];
            auto d2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false); hi = d2[0] | d2[1];
            accw = ((unsigned long long)hi << 32) | lo;
          }
          if (owner && quarter == 0) R |= accw;
          NMS_BACC(6);
        }
      }
      if (lane == 0) { ctl[0] = nk; ctl[1] = total; ctl[2] = done ? 1 : 0; }
    }
    __syncthreads();
    NMS_ACC(2);   // B: serial walk (C_far of the previous super-chunk beside it)
    const int nk = ctl[0];
    const bool done = ctl[2] != 0;
    if (keep) {   // the super-chunk's survivors into the keep list, by everybody
      const int first = ctl[1] - nk;
      for (int i = tid; i < nk; i += kSweepThreads) keep[first + i] = base + kept_list[i];
    }
    // ---- C_near: the kept rows over the 16 words of the next super-chunk, everybody
    const int wnext = w0 + nw;
    const int near_hi = min(col_end, wnext + kSuperWords);
    if (!done) or_rows(kept_list, nk, base, wnext, near_hi, tid, kSweepThreads);
    far_nk = done ? 0 : nk; far_base = base; far_lo = near_hi;
    cur ^= 1;
    __syncthreads();
    NMS_ACC(3);   // C_near
    if (done) break;
  }
  // the last super-chunk's far columns (none when the loop ran to the end of the phase's columns; kept general)
  if (ctl[2] == 0 && far_nk > 0 && far_lo < col_end) {
    or_rows(kept_buf[cur ^ 1], far_nk, far_base, far_lo, col_end, tid, kSweepThreads);
    __syncthreads();
  }
  const int total = ctl[1];
  if (two_phase) {
    const bool finished = ctl[2] != 0 || sc_last >= n_super;
    if (sc_begin == 0) {
      if (!finished) {   // park the state for the second phase
        for (int j = tid; j < col_blocks; j += kSweepThreads) state[2 + j] = remv[j];
        if (tid == 0) { state[1] = (unsigned long long)total; state[0] = 0ULL; }
        return;
      }
      if (tid == 0) state[0] = 1ULL;
    }
  }
  if (tid == 0 && num_out) num_out[img] = total;
  // ---- optional epilogue: RoI rows [img, x1, y1, x2, y2], zero padded (proposal_layer.py:157-159)
  if (rois_out) {
    __syncthreads();
    const float* b = boxes + img * box_batch_stride;
    float* out = rois_out + (long)img * rois_rows * 5;
    for (int r = tid; r < rois_rows; r += kSweepThreads) {
      float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
      if (r < total) {
        const float* p = b + (long)keep[r] * boxes_dim;
        x1 = p[0]; y1 = p[1]; x2 = p[2]; y2 = p[3];
      }
      out[r * 5 + 0] = (float)img;
      out[r * 5 + 1] = x1;
      out[r * 5 + 2] = y1;
      out[r * 5 + 3] = x2;
      out[r * 5 + 4] = y2;
    }
  }
}

}  // namespace
#ifdef DTT_NMS_TRACE
extern "C" int dtt_nms_cycles_read(unsigned long long* host, int reset) {
  (void)hipDeviceSynchronize();
  int ok = hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_nms_cycles), sizeof(unsigned long long) * 64 * 8) == hipSuccess;
  if (reset) { static unsigned long long zero[64 * 8]; ok &= hipMemcpyToSymbol(HIP_SYMBOL(dtt_nms_cycles), zero, sizeof(zero)) == hipSuccess; }
  return ok;
}
#endif
namespace {

size_t sweep_lds_bytes(int col_blocks) {
  return (size_t)kSuper * kRowStride * 8 + (size_t)col_blocks * 8 + (size_t)kSuper * 8 + (size_t)2 * kSuper * 4 + 16;
}

}  // namespace

size_t dtt_nms_mask_bytes(int boxes_num) {
  const long cb = (boxes_num + kTile - 1) / kTile;
  // the bit matrix + the parked sweep state of a two-phase run (done flag, survivor count, removal words)
  const size_t n = boxes_num > 0 ? boxes_num : 1, c = cb > 0 ? cb : 1;
  return (n * c + c + kStateWords + n) * sizeof(unsigned long long);
}

// Two phases when few survivors are wanted (the proposal layer keeps 300 of 6000): the sweep visits boxes in score order
// and stops at max_keep, so with a keep rate around 40 % it never looks past the first ~750 boxes.  Phase 1 = the bit matrix
// of the first ceil(3 * max_keep / 1024) super-chunks AGAINST THEMSELVES (rows and columns below R1: 136 tiles per image
// instead of 4465 at 6000 boxes) + their sweep; phase 2 = every other tile, the kept rows replayed over the new columns,
// and the rest of the sweep -- both of whose launches return at once for images phase 1 finished.  Same keep list either way
// (the sweep is the same recursion).  Returns the number of super-chunks in phase 1 (0 = single phase).
int dtt_nms_split(int n_max, int max_keep, int have_keep_out) {
  static const bool single_phase_only = getenv("DTT_NMS_SINGLE_PHASE") != nullptr;   // developer A/B switch
  const int n_super = (n_max + kSuper - 1) / kSuper;
  if (max_keep > 0 && have_keep_out && !single_phase_only) {
    const int r1 = (int)((3L * max_keep + kSuper - 1) / kSuper);
    if (2 * r1 <= n_super) return r1;
  }
  return 0;
}

static int nms_prepare(int n_max, int batch, int boxes_dim, size_t* lds) {
  const int cb = (n_max + kTile - 1) / kTile;
  DTT_REQUIRE(n_max > 0 && batch > 0, "nms: empty problem (n=%d, batch=%d)", n_max, batch);
  DTT_REQUIRE(boxes_dim >= 4, "nms: boxes_dim must be >= 4 (got %d)", boxes_dim);
  *lds = sweep_lds_bytes(cb);
  DTT_REQUIRE(*lds <= 160 * 1024, "nms: %d boxes exceed the LDS-resident sweep state (%zu B)", n_max, *lds);
  static DttDeviceOnce attr_set_once;
  bool& attr_set = attr_set_once.here();   // the attribute is per device, not per process
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { dtt_set_error("nms: cannot raise dynamic LDS limit: %s", hipGetErrorString(e)); return 0; }
    attr_set = true;
  }
  return 1;
}

// Phase 1 (split > 0: needs the first split * 1024 boxes only) or the whole NMS (split == 0).
int dtt_nms_phase1(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image, int n_max, int batch,
                   float thresh, int max_keep, unsigned long long* mask, long mask_batch_stride, int* keep_out,
                   long keep_batch_stride, int* num_out, float* rois_out, int rois_rows, int split, hipStream_t stream) {
  size_t lds;
  if (!nms_prepare(n_max, batch, boxes_dim, &lds)) return 0;
  const int cb = (n_max + kTile - 1) / kTile;
  const int n_super = (n_max + kSuper - 1) / kSuper;
  const int rb1 = split ? min(cb, split * kSuperWords) : cb;   // row AND column blocks of the first mask launch
  dtt_prof_begin("nms_mask", stream);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(rb1, rb1, batch), dim3(kMaskThreads), 0, stream, boxes, boxes_dim, box_batch_stride,
                     n_per_image, n_max, thresh, mask, mask_batch_stride, cb, 0, rb1, 0, 0, nullptr, 0L, 0);
  dtt_prof_end("nms_mask", stream);
  DTT_CHECK_LAUNCH("nms_mask_kernel");
  dtt_prof_begin("nms_sweep", stream);
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(batch), dim3(kSweepThreads), lds, stream, mask, mask_batch_stride,
                     n_per_image, n_max, cb, max_keep, keep_out, keep_batch_stride, num_out, boxes, boxes_dim,
                     box_batch_stride, rois_out, rois_rows, 0, split ? split : n_super, split ? 1 : 0);
  dtt_prof_end("nms_sweep", stream);
  DTT_CHECK_LAUNCH("nms_sweep_kernel");
  return 1;
}

// Phase 2 (split > 0 only; needs every box): the tiles phase 1 skipped and the rest of the sweep.
int dtt_nms_phase2(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image, int n_max, int batch,
                   float thresh, int max_keep, unsigned long long* mask, long mask_batch_stride, int* keep_out,
                   long keep_batch_stride, int* num_out, float* rois_out, int rois_rows, int split, hipStream_t stream) {
  if (!split) return 1;
  size_t lds;
  if (!nms_prepare(n_max, batch, boxes_dim, &lds)) return 0;
  const int cb = (n_max + kTile - 1) / kTile;
  const int n_super = (n_max + kSuper - 1) / kSuper;
  const int rb1 = min(cb, split * kSuperWords);
  if (cb > rb1) {   // column blocks rb1 .. cb-1, all their rows (a short grid that loops over the row blocks)
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb - rb1, 8, batch), dim3(kMaskThreads), 0, stream, boxes, boxes_dim, box_batch_stride,
                       n_per_image, n_max, thresh, mask, mask_batch_stride, cb, 0, cb, rb1, 1, keep_out, keep_batch_stride,
                       rb1 * kTile);
    DTT_CHECK_LAUNCH("nms_mask_kernel (phase 2)");
  }
  hipLaunchKernelGGL(nms_sweep_kernel, dim3(batch), dim3(kSweepThreads), lds, stream, mask, mask_batch_stride,
                     n_per_image, n_max, cb, max_keep, keep_out, keep_batch_stride, num_out, boxes, boxes_dim,
                     box_batch_stride, rois_out, rois_rows, split, n_super, 1);
  DTT_CHECK_LAUNCH("nms_sweep_kernel (phase 2)");
  return 1;
}

int dtt_nms_batched_launch(const float* boxes, int boxes_dim, long box_batch_stride, const int* n_per_image,
                           int n_max, int batch, float thresh, int max_keep, unsigned long long* mask,
                           long mask_batch_stride, int* keep_out, long keep_batch_stride, int* num_out,
                           float* rois_out, int rois_rows, hipStream_t stream) {
  const int split = dtt_nms_split(n_max, max_keep, keep_out != nullptr);
  dtt_prof_begin("nms_op", stream);   // (event tag: the whole NMS, both phases)
  if (!dtt_nms_phase1(boxes, boxes_dim, box_batch_stride, n_per_image, n_max, batch, thresh, max_keep, mask, mask_batch_stride,
                      keep_out, keep_batch_stride, num_out, rois_out, rois_rows, split, stream))
    return 0;
  const int ok = dtt_nms_phase2(boxes, boxes_dim, box_batch_stride, n_per_image, n_max, batch, thresh, max_keep, mask, mask_batch_stride,
                                keep_out, keep_batch_stride, num_out, rois_out, rois_rows, split, stream);
  dtt_prof_end("nms_op", stream);
  return ok;
}

extern "C" size_t dtt_nms_workspace_bytes(int boxes_num) { return dtt_nms_mask_bytes(boxes_num); }

extern "C" int dtt_nms(int* keep_out, int* num_out, const float* boxes, int boxes_num, int boxes_dim,
                       float nms_overlap_thresh, int max_keep, void* workspace, size_t workspace_bytes,
                       void* stream) {
  DTT_REQUIRE(keep_out && num_out && boxes, "nms: null pointer");
  DTT_REQUIRE(boxes_num > 0, "nms: boxes_num must be > 0 (the wrapper returns [] for empty input, nms_wrapper.py:13-14)");
  DTT_REQUIRE(workspace && workspace_bytes >= dtt_nms_mask_bytes(boxes_num),
              "nms: workspace too small (%zu < %zu)", workspace_bytes, dtt_nms_mask_bytes(boxes_num));
  return dtt_nms_batched_launch(boxes, boxes_dim, 0, nullptr, boxes_num, 1, nms_overlap_thresh, max_keep,
                                static_cast<unsigned long long*>(workspace), 0, keep_out, 0, num_out, nullptr, 0,
                                static_cast<hipStream_t>(stream));
}

// Cross-frame correlation, BACKWARD on channels-last maps, band-stationary / halo-streamed form (gfx950): both gradients of
// Correlation_backward_input1 / _input2 (correlation/src/correlation_cuda_kernel.cu:108-290) for kernel_size 1,
// stride1 == stride2, max_displacement / stride <= 8, channels % 64 == 0 -- the three correlations D&T trains (rfcn.py:58-60).
//
//   With G[p, q] = gradOut[p, q - p] (p: output pixel, q: displaced pixel, zero outside the window or the image)
//       gradInput1[p, c] = 1/C * sum_q G[p, q] * f2[q, c]          gradInput2[q, c] = 1/C * sum_p G[p, q] * f1[p, c]
//   i.e. for a 4 x 4 block of TARGET pixels a [16 targets x 4 halo pixels] x [4 halo pixels x 16 channel quads] product per
//   step of the exact-f32 v_mfma_f32_16x16x4_f32, the reduction running over the (4 + 2R)^2 halo of the OTHER frame.
//   The band does not depend on the channel: a wave keeps the NBR^2 * 4 band words of its block in registers (MFMA A
//   operand) for its whole life.  The contraction runs over PIXELS, so one ds_read_b128 of a lane (4 consecutive channels of
//   one halo pixel) feeds FOUR MFMAs when a chunk is 64 channels wide -- each MFMA takes the channels = s (mod 4) as its 16
//   columns -- and the lane then holds, per target pixel, four consecutive channels: the result leaves as float4 stores
//   straight from the accumulators.  No LDS epilogue, no exchange, no partial sums, every gradient element written once.
//
//   * Workgroup = 8 compute waves = a tile of th x tw target blocks (2 x 4; 4 x 1 / 2 x 1 ... along odd map edges) + 4 loader
//     waves (kComp / kLoad below) that issue the LDS-DMA (global_load_lds_dwordx4, scalar base) and keep the ring's bookkeeping
//     off the compute waves: three waves per SIMD at 168 registers each (band 100 + accumulators 16 + operands).
//   * The halo of a 64-channel group is streamed as BLOCK ROWS (4 halo rows x the tile's halo width x 64 channels = one ring
//     slot, 32 KB for a 2 x 4 tile): wave row wy consumes block rows wy .. wy + NBR - 1 of a group, i.e. it runs `wy` ring
//     positions ahead of wave row 0 -- every wave has NBR steps of work per group although the tile's halo has th + NBR - 1
//     block rows (a lock-step schedule would idle (th - 1) / (th + NBR - 1) of the matrix pipe).  One barrier per step
//     (80 MFMAs per wave at R = 8).
//   * Work items = (tile, run of channel groups), laid out by a host-side plan: the chunk length is chosen by simulating the
//     dispatch (longest first) over the CUs; the item order rides in the kernel arguments so that every XCD gets its share of
//     long and short items and the items that are resident together on an XCD are neighbouring tiles working on the SAME
//     channel groups (their halos overlap: L2 serves the overlap).
//   * The band words are laid out once per op by a small kernel (corr_bwd_band_kernel) in the caller's workspace, in MFMA
//     register order [direction][image][block][entry][lane] with the validity (window, image borders, output range) folded
//     in -- gradOut may lie as the reference's (n, D*D, oh, ow) planes or as columns of position-major rows.  A wave then
//     fetches its band with NBR^2 * 4 fully coalesced loads while the first ring positions are already in flight.  (Gathering
//     the band inside the main kernel -- directly or staged through LDS -- costs 1.5 - 3 k vector instructions per wave on
//     the critical path of every workgroup: first version of this file.)
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGC = 64;                       // channels per group
constexpr int kComp = 8, kLoad = 4, kThreads = (kComp + kLoad) * 64;   // 2 compute waves + 1 loader per SIMD (168 registers each)
constexpr int kLdsMax = 144 * 1024;           // 16 KB of the CU's 160 stay free for small kernels of other streams
constexpr int kMaxNI = 12;                    // DMA instructions per loader per ring position (halo width <= 48 pixels)
constexpr int kMaxSeg = 8;
constexpr int kTable = 1536;                  // work-item order in the kernel arguments (16-bit item numbers)

struct BSeg { int tile0, by0, bx0, nty, ntx, th, tw, ahead; };   // ahead: ring slots beyond the th being read

struct BGeom {
  const float* other[2];               // the frame the window runs over (frame t+tau for gradInput1, frame t for gradInput2), channels-last;
  float* grad[2];                      // gradient of the target frame, channels-last.  [1]: the second direction of a MERGED launch
  const float* band;                   // band words [quarter][direction][image][block row][block column][NBR^2][64 lanes][4 steps]
  int nimg;                            // images per direction: a merged launch plans 2 * nimg "images", image n >= nimg = direction 1's n - nimg
  int gh, gw;                          // its grid of 4 x 4 target blocks
  long sb;                             // floats between images of other / grad
  unsigned sy4, sx4;                   // bytes between vertically / horizontally adjacent lattice pixels
  int C, H, W;                         // channels, lattice size
  int stride, ih, iw;                  // lattice stride, image size; cell_fill: a target also zeroes the stride x stride cell of
  int cell_fill;                       // non-lattice pixels behind it (every pixel is then written by the kernel: no memset)
  unsigned py4, px4;                   // bytes between vertically / horizontally adjacent IMAGE pixels
  int oh, ow, origin;                  // output size; output (y, x) <-> lattice pixel (origin + y, origin + x)
  int R, D, D2;
  unsigned d2_magic;                   // 2^32 / D2 + 1
  int lo_y, lo_x, hi_y, hi_x;          // targets, in output coordinates (inclusive)
  float inv;                           // 1 / C
  int nseg, tiles_per_image, tiles_total;
  BSeg seg[kMaxSeg];
  int chunk, rem, ngroups;             // item -> (chunk index ci = item / tiles_total, tile = item % tiles_total); the channel groups are
  int use_table;                       // dealt evenly: the first `rem` chunks hold chunk + 1 groups, the others `chunk`
  // window radius > 8 (MULTI): the (up to) 9 x 9 window blocks are walked as 2 x 2 QUARTERS of NBR x NBR blocks inside one launch --
  // per channel group the halo of quarter (qa, qb) starts 4 NBR (qa, qb) pixels further on, the accumulators run through all four,
  int nyl, nxl;                        // the second quarter row / column holds only nyl / nxl blocks that lie inside the window,
  long band_q;                         // and a quarter's band words lie band_q floats behind the previous quarter's
  int ablate;                          // developer timing experiments (DTT_CORR_BWD_ABLATE): 1 no DMA, 2 no MFMA, 4 no stores, 8 no band loads
  unsigned short table[kTable];
};
static_assert(sizeof(BGeom) <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ void dma16b(const char* sbase, unsigned voff, unsigned lds_addr) {
  // m0 (the DMA's LDS base) is put back inside the statement: it is a reserved register the compiler neither allocates nor saves
  // around inline asm -- a statement that merely listed it as clobbered would rely on hipcc never keeping a value of its own there
  unsigned keep_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep_m0)
               : "s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory");
}
__device__ __forceinline__ const char* uptrb(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  return (const char*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ void wait_vmcnt_b(int n) {   // s_waitcnt vmcnt(n), n wave-uniform at run time
#define DTT_W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    DTT_W1(0) DTT_W1(1) DTT_W1(2) DTT_W1(3) DTT_W1(4) DTT_W1(5) DTT_W1(6) DTT_W1(7) DTT_W1(8) DTT_W1(9) DTT_W1(10) DTT_W1(11)
    DTT_W1(12) DTT_W1(13) DTT_W1(14) DTT_W1(15) DTT_W1(16) DTT_W1(17) DTT_W1(18) DTT_W1(19) DTT_W1(20) DTT_W1(21) DTT_W1(22)
    DTT_W1(23) DTT_W1(24)
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;   // (a stricter wait than asked, never a looser one)
  }
#undef DTT_W1
}
__device__ __forceinline__ void wg_barrier_b() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__host__ __device__ __forceinline__ int mdiv32b(int n, unsigned magic) {
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n;
}

// s_waitcnt lgkmcnt(LEFT) that names a burst's registers as read-write operands: all but the LEFT youngest LDS reads have
// returned, and nothing that consumes those registers can be scheduled in front of the wait
template <int LEFT>
__device__ __forceinline__ void landed5(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(LEFT));
}
template <int LEFT>
__device__ __forceinline__ void landed3(f32x4& a, f32x4& b, f32x4& c) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(LEFT));
}

struct BItem { int n, th, tw, ahead, Y0, X0, g0, ng; };   // image, tile shape (blocks), ring depth, tile origin (output coordinates), channel groups

__host__ __device__ __forceinline__ BItem bw_decode(const BGeom& g, int item) {
  BItem it;
  const int ci = item / g.tiles_total, tile = item - ci * g.tiles_total;
  it.n = tile / g.tiles_per_image;
  const int r = tile - it.n * g.tiles_per_image;
  BSeg sg = g.seg[0];
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < g.nseg && r >= g.seg[i].tile0) sg = g.seg[i];
  const int local = r - sg.tile0;
  const int tyi = local / sg.ntx, txi = local - tyi * sg.ntx;
  it.th = sg.th; it.tw = sg.tw; it.ahead = sg.ahead;
  it.Y0 = g.lo_y + 4 * (sg.by0 + tyi * sg.th);
  it.X0 = g.lo_x + 4 * (sg.bx0 + txi * sg.tw);
  it.g0 = ci * g.chunk + min(ci, g.rem);
  it.ng = g.chunk + (ci < g.rem ? 1 : 0);
  return it;
}

template <int NBR, bool MULTI = false>
__global__ __launch_bounds__(kThreads) void corr_bwd_stream_kernel(BGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NB2 = NBR * NBR;
  constexpr int NSEG = MULTI ? 4 : 1;                 // window quarters walked per channel group
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  const int item = g.use_table ? (int)g.table[bid] : dtt_xcd_remap(bid, gridDim.x);
  BItem it = bw_decode(g, item);
  const int dir = it.n >= g.nimg ? 1 : 0;             // (merged launch: the second half of the planned images is direction 1)
  const int n_eff = it.n;
  it.n -= dir ? g.nimg : 0;
  const int th = it.th, tw = it.tw, nwv = th * tw;
  const int HC = 4 * (tw + NBR - 1);                  // halo width of the tile, pixels
  const int slot_bytes = 4 * HC * kGC * 4;            // one halo block row of one channel group
  const int S = th + it.ahead;                        // ring slots
  const int nyl = MULTI ? g.nyl : NBR, nxl = MULTI ? g.nxl : NBR;
  // halo block rows per channel group: th + ny - 1 per quarter
  const int rows_group = MULTI ? 2 * (th + NBR - 1) + 2 * (th + nyl - 1) : th + NBR - 1;
  const int npos = it.ng * rows_group;
  const unsigned lds0 = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)lds;
  const long img = (long)it.n * g.sb;
  // A tall tile at a segment boundary (next channel group / next quarter): th rows step into fresh positions at once and the ring
  // holds fewer than 2 th slots, so the slots only come free at the barrier -- such a step has two barriers (fill, drain, meet
  // again).  Every wave derives this from the tile shape alone.
  const bool tall = th > it.ahead;

  if (wave >= kComp) {
    // ================================================================ loaders: SALU + LDS-DMA only
    // Instruction j = lw + 4 i of a position fills slot pixels [4 j, 4 j + 4) (x 64 channels = 1 KB): halo row (4 j) / HC of
    // the block row, columns (4 j) % HC + lane / 16.  All bookkeeping of the ring (what has landed, what may be refilled) lives
    // here: the compute waves -- which share their SIMDs with these -- see a barrier per step and nothing else.  (First
    // version: every wave issued its share of the DMA; the few dozen scalar instructions per step, executed by all eight
    // waves at once right behind the barrier, added their full length to every step: a wave issues one instruction per
    // ~4 cycles, and nobody was feeding the matrix pipe meanwhile -- kernel time = MFMA time + 27 us of bookkeeping.)
    const int lw = wave - kComp;
    __builtin_amdgcn_s_setprio(3);
    unsigned voff[kMaxNI], voff_b[MULTI ? kMaxNI : 1];   // voff_b: the quarters of the second column (halo 4 NBR pixels further right)
    int row_of[kMaxNI];
    int ni = 0;
#pragma unroll
    for (int i = 0; i < kMaxNI; ++i) {
      const int j = lw + kLoad * i;
      voff[i] = 0; row_of[i] = 0;
      if (MULTI) voff_b[i] = 0;
      if (j < HC) {
        const int t = (4 * j) / HC, c = 4 * j - t * HC + (lane >> 4);
        const int x = min(max(g.origin + it.X0 - g.R + c, 0), g.W - 1);   // out-of-image pixels: any in-bounds address (their band words are zero)
        voff[i] = (unsigned)x * g.sx4 + (unsigned)((lane & 15) << 4);
        if (MULTI) {
          const int xb = min(max(g.origin + it.X0 - g.R + 4 * NBR + c, 0), g.W - 1);
          voff_b[i] = (unsigned)xb * g.sx4 + (unsigned)((lane & 15) << 4);
        }
        row_of[i] = t;
        ++ni;
      }
    }
    if (g.ablate & 1) ni = 0;
    const char* obase = reinterpret_cast<const char*>(g.other[dir] + img) + (long)it.g0 * (kGC * 4);
    // the next position to issue, tracked incrementally (all scalar): its group's base address, quarter, halo block row, ring slot
    int issued = 0, i_hr = 0, i_slot = 0, i_q = 0, i_nh = th + NBR - 1;
    const int y_first = g.origin + it.Y0 - g.R;
    auto issue_next = [&]() {
      const unsigned dst = lds0 + (unsigned)(i_slot * slot_bytes + lw * 1024);
      const int y_q = y_first + (MULTI && (i_q & 2) ? 4 * NBR : 0) + 4 * i_hr;
      const bool right = MULTI && (i_q & 1);
#pragma unroll
      for (int i = 0; i < kMaxNI; ++i)
        if (i < ni) {
          const int y = min(max(y_q + row_of[i], 0), g.H - 1);
          dma16b(uptrb(obase + (unsigned long long)((unsigned)y * g.sy4)), right ? voff_b[MULTI ? i : 0] : voff[i], dst + (unsigned)(i * kLoad * 1024));
        }
      ++issued;
      if (++i_slot == S) i_slot = 0;
      if (++i_hr == i_nh) {
        i_hr = 0;
        if (MULTI) {
          if (++i_q == NSEG) { i_q = 0; obase += kGC * 4; }
          i_nh = th + ((i_q & 2) ? nyl : NBR) - 1;
        } else {
          obase += kGC * 4;
        }
      }
    };
    while (issued < min(S, npos)) issue_next();        // the ring is filled first (positions 0 .. th - 1 are read in step 0)
    int base = 0;
    for (int gi = 0; gi < it.ng; ++gi) {
      for (int q = 0; q < NSEG; ++q) {
        const int ny = (q & 2) ? nyl : NBR;
        for (int qi = 0; qi < ny; ++qi, ++base) {
          // Positions <= base + th - 1 are read in this step.  After the barrier the ring is refilled as far as it goes: position
          // P's slot is free once everybody is done with position P - S < base.
          const int need = min(base + th - 1, npos - 1);
          const int fill = min(base + S - 1, npos - 1);
          if (tall && qi == 0 && (gi | q) != 0) {
            wg_barrier_b();
            while (issued <= fill) issue_next();
            wait_vmcnt_b(0);
            wg_barrier_b();
          } else {
            wait_vmcnt_b((issued - 1 - need) * ni);   // mine of the later positions may still fly (loads return in order)
            wg_barrier_b();                           // everybody's share has landed; everybody is done with the positions below `base`
            while (issued <= fill) issue_next();
          }
        }
        base += th - 1;                               // (the next segment's first block row)
      }
    }
    return;
  }

  // ================================================================ compute
  const bool active = wave < nwv;
  const int wy = active ? wave / tw : 0, wx = active ? wave - wy * tw : 0;
  // ---------------------------------------------------------------- band: NBR^2 x 4 words per lane, in register order (one 16-byte
  // load per window block: the four steps of a block are consecutive)
  f32x4 band[NB2];
  // (waves without a block -- a short tile -- read block (0, 0)'s words and never use them; ablation 8 reads ONE block's words
  //  everywhere: the loads stay, their traffic goes)
  const int bby = (g.ablate & 8) ? 0 : (it.Y0 - g.lo_y) / 4 + wy, bbx = (g.ablate & 8) ? 0 : (it.X0 - g.lo_x) / 4 + wx;
  const char* bpu = uptrb(reinterpret_cast<const char*>(g.band + ((((long)((g.ablate & 8) ? 0 : n_eff) * g.gh + bby) * g.gw + bbx) * NB2) * 256));
  const unsigned lane16 = (unsigned)lane * 16u;
  // MULTI: the band loads are inline asm the compiler does not track (a tracked load that is re-issued inside the loop drags counted
  // waits into every MFMA group, and a conditional one a branch and a full drain per load): row qi of a quarter's words is requested
  // right behind step qi of the quarter before, 5 x 16 bytes per lane, and released by ONE counted wait in front of step qi
  auto band_row = [&](const char* base, int qi) {
#pragma unroll
    for (int qj = 0; qj < NBR; ++qj) {
      const int e = qi * NBR + qj;
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(band[e]) : "v"(lane16), "s"(base + (e / 4) * 4096), "n"((e % 4) * 1024));
    }
  };
  if constexpr (MULTI) {
#pragma unroll
    for (int qi = 0; qi < NBR; ++qi) band_row(bpu, qi);
  } else {
    const f32x4* bp = reinterpret_cast<const f32x4*>(bpu) + lane;
#pragma unroll
    for (int b = 0; b < NB2; ++b) band[b] = bp[b * 64];
  }
  const long band_qb = MULTI ? g.band_q * 4 : 0;      // bytes between the quarters' band words
  // ---------------------------------------------------------------- store descriptors: lane -> target row tyi = lane / 16, columns r = 0 .. 3
  // (two registers: the byte offset of column 0 and a mask of the columns that are targets; the rest is recomputed per group)
  unsigned st_base = 0, st_mask = 0;
  {
    const int ty = it.Y0 + 4 * wy + (lane >> 4), ly = ty + g.origin;
    const int lx0 = it.X0 + 4 * wx + g.origin;
    const bool row_ok = active && ty >= g.lo_y && ty <= g.hi_y && ly >= 0 && ly < g.H && !(g.ablate & 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tx = it.X0 + 4 * wx + r, lx = lx0 + r;
      if (row_ok && tx >= g.lo_x && tx <= g.hi_x && lx >= 0 && lx < g.W) st_mask |= 1u << r;
    }
    st_base = (unsigned)max(ly, 0) * g.sy4 + (unsigned)max(lx0, 0) * g.sx4 + (unsigned)((lane & 15) << 4);
    if (lx0 < 0) st_base -= (unsigned)(-lx0) * g.sx4;       // (columns left of the image are masked out; the offset stays consistent)
  }
  char* gbase = reinterpret_cast<char*>(g.grad[dir] + img) + (long)it.g0 * (kGC * 4);

  // ---------------------------------------------------------------- main loop: one step = one halo block row of one (group, quarter)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const unsigned rd_lane = lds0 + (unsigned)(wx * 1024 + lane * 16);
  const unsigned row = (unsigned)(HC * kGC * 4);
  const bool do_mfma = active && !(g.ablate & 2);
  // Operand reads are inline asm the compiler does not track, released by ONE hand-counted s_waitcnt per burst (LDS returns in
  // order): hipcc's own bookkeeping puts a counted wait in front of every group of four MFMAs, and stray issue slots inside
  // an MFMA stream cost the pipe tens of cycles each.  The wait lists the burst's registers as read-write operands, so the MFMAs
  // that consume them cannot be moved in front of it.
  f32x4 bv[2][NBR];
  auto rd = [&](unsigned addr, int buf) {
#pragma unroll
    for (int qj = 0; qj < NBR; ++qj)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bv[buf][qj]) : "v"(addr), "n"(qj * 4 * kGC * 4));
  };
#define DTT_LANDED(buf, LEFT)                                                                                          \
  do {                                                                                                                \
    if constexpr (NBR == 5) landed5<LEFT>(bv[buf][0], bv[buf][1], bv[buf][2], bv[buf][NBR - 2], bv[buf][NBR - 1]);      \
    else landed3<LEFT>(bv[buf][0], bv[buf][1], bv[buf][2]);                                                            \
  } while (0)
  int r_slot = wy % S;                                // ring slot of the position this wave reads: wave row wy runs wy positions ahead
  // Measured and dropped (A/B builds on one box, conv5 / conv4 per gradient): a wave row that is not the tile's last requesting
  // the next step's first burst BEFORE the barrier -- its position has landed already -- so that it starts the step with MFMAs.
  // As a run-time flag inside the loop: 44 registers spilled (two register states merge at every step).  As a second copy of the
  // loop chosen once per wave: no spills in the steps, 93.4 / 52.2 us against 91.9 / 50.8 without -- the 24 bytes of scratch it
  // does need are reloaded in the first step of every group.
  {
    for (int gi = 0; gi < it.ng; ++gi) {
      for (int q = 0; q < NSEG; ++q) {
        // MULTI: quarter q = (qa, qb) of the window holds ny x nx blocks that can be non-zero; the rest is skipped on both sides
        // of the ring (no step, no DMA).  The band words of the NEXT quarter replace this one's block row by block row, each
        // right behind the step that consumed it -- five 16-byte loads per step that have the rest of the quarter to land.
        const int ny = (MULTI && (q & 2)) ? nyl : NBR, nx = (MULTI && (q & 1)) ? nxl : NBR;
        // (the last quarter of the item requests quarter 0's words once more: in bounds, never used -- no condition on any load)
        const char* bn = bpu + (long)((q + 1) & 3) * band_qb;
#pragma unroll
        for (int qi = 0; qi < NBR; ++qi) {
          if (!MULTI || qi < ny) {
            // nothing of this wave's is outstanding at a barrier that the compiler knows of: band loads (first step) and stores
            // (later groups) were issued long before -- the builtin tells it so, and its own waits stay out of the loop
            if (!MULTI && qi == 0 && gi == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the band has arrived
            if (tall && qi == 0 && (gi | q) != 0) wg_barrier_b();
            wg_barrier_b();
            if constexpr (MULTI) {
              // Row qi's words: (NBR - 1 - qi) rows were requested behind them in the quarter before and qi rows in this one -- 20 loads
              // may still fly; 16 leaves room for the four stores of a finished group should they retire out of order with the loads.
              static_assert(NBR == 5, "the counted wait below is written for 5 x 5 quarters");
              asm volatile("s_waitcnt vmcnt(16)" : "+v"(band[qi * NBR]), "+v"(band[qi * NBR + 1]), "+v"(band[qi * NBR + 2]),
                                                   "+v"(band[qi * NBR + 3]), "+v"(band[qi * NBR + 4]));
            }
            if (do_mfma) {
              const unsigned sp = rd_lane + (unsigned)(r_slot * slot_bytes);
              auto mm = [&](int t, int buf) {
#pragma unroll
                for (int qj = 0; qj < NBR; ++qj) {
                  if (MULTI && qj >= nx) continue;
                  const float a = band[qi * NBR + qj][t];
                  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[buf][qj][0], acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[buf][qj][1], acc1, 0, 0, 0);
                  acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[buf][qj][2], acc2, 0, 0, 0);
                  acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[buf][qj][3], acc3, 0, 0, 0);
                }
              };
              // one burst of NBR ds_read_b128 per halo row t (4 MFMAs per read), issued one row ahead of the MFMAs that consume it
              __builtin_amdgcn_sched_barrier(0);
              rd(sp, 0); rd(sp + row, 1);
              DTT_LANDED(0, NBR); mm(0, 0);
              __builtin_amdgcn_sched_barrier(0);
              rd(sp + 2 * row, 0);
              DTT_LANDED(1, NBR); mm(1, 1);
              __builtin_amdgcn_sched_barrier(0);
              rd(sp + 3 * row, 1);
              DTT_LANDED(0, NBR); mm(2, 0);
              __builtin_amdgcn_sched_barrier(0);
              DTT_LANDED(1, 0); mm(3, 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (++r_slot == S) r_slot = 0;
          }
          if constexpr (MULTI) band_row(bn, qi);
        }
        r_slot += th - 1;                               // the next segment starts th block rows further on
        if (r_slot >= S) r_slot -= S;
      }
      // the group is complete: D[m = 4 * (lane / 16) + r][n = lane % 16] of accumulator s is channel 4 n + s of target pixel m
      char* dst = gbase + (long)gi * (kGC * 4) + st_base;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (st_mask & (1u << r)) {
          f32x4 o = {acc0[r] * g.inv, acc1[r] * g.inv, acc2[r] * g.inv, acc3[r] * g.inv};
          char* px = dst + (unsigned)r * g.sx4;
          *reinterpret_cast<f32x4*>(px) = o;
          if (g.cell_fill) {   // strided lattice (conv3): the image pixels between the lattice points have no gradient
            const int ly = it.Y0 + 4 * wy + (lane >> 4) + g.origin, lx = it.X0 + 4 * wx + r + g.origin;
            const int ch = min(g.stride, g.ih - ly * g.stride), cw = min(g.stride, g.iw - lx * g.stride);
            for (int cy = 0; cy < ch; ++cy)
              for (int cx = 0; cx < cw; ++cx)
                if (cy | cx) *reinterpret_cast<f32x4*>(px + (unsigned)cy * g.py4 + (unsigned)cx * g.px4) = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      acc0 = acc1 = acc2 = acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}
#undef DTT_LANDED

// ------------------------------------------------------------------------------------------------ the band, once per op
// band[quarter][dir][n][by][bx][qi * NBR + qj][lane][t]: what lane (m = lane % 16: target pixel (m / 4, m % 4) of block (by, bx);
// k = lane / 16) of the wave that owns the block feeds the MFMA of window block (qi, qj), step t as its A operand = the gradient
// that couples target m with the other frame's halo pixel (4 qi + t, 4 qj + k) of the block's halo -- zero where the pair lies
// outside the window, the output range or the image.  The four steps of a block are one 16-byte piece per lane: the wave that owns the
// block fetches a window block's words with ONE global_load_dwordx4.  One workgroup per (quarter, direction, image, block): the block's
// 16 x D*D pairs are staged through LDS with loads that follow gradOut's contiguous axis (the first version gathered straight from
// memory: 25 loads per thread each touching 64 cache lines in the planes layout, 16.6 us per conv5 op).
struct BandGeom {
  const float* gout; long g_sb, g_sc, g_sp;
  float* band;
  int oh, ow, origin, H, W, R, D;
  unsigned d_magic, d2_magic;          // 65536 / D + 1 (exact for n < 4096), 2^32 / (D * D) + 1
  int nq, blocks_q;                    // window radius > 8: nq x nq quarters of NBR x NBR window blocks, blocks_q workgroups each,
  long band_q;                         // band_q floats of band words each
  int ablate;                          // developer timing experiments (DTT_CORR_BWD_ABLATE): 32 no gradOut loads, 64 no band stores
  int lo_y[2], lo_x[2], gh[2], gw[2];
  long dir_off[2];                     // floats from a quarter's first word to a direction's words
  int batch;
};

template <int NBR, bool SUB>
__global__ __launch_bounds__(256) void corr_bwd_band_kernel(BandGeom g) {
  constexpr int NB2 = NBR * NBR;
  // the displacements one workgroup can touch per axis: the whole window (2 R + 1 <= 4 (NBR - 1) + 1), or -- SUB: window radius > 8,
  // the window covered in quarters of NBR x NBR blocks -- the 4 NBR + 3 rows a quarter's halo spans around a 4 x 4 block
  constexpr int WIN = SUB ? 4 * NBR + 3 : 4 * (NBR - 1) + 1, WIN2 = WIN * WIN;
  constexpr int NIT = (16 * WIN2 + 255) / 256;
  __shared__ float G[16 * (WIN2 + 1)];                                // G[m * ldg + w]: gradOut of the pair (target m, displacement w)
  int blk = blockIdx.x;
  const int quarter = SUB ? blk / g.blocks_q : 0;
  blk -= quarter * g.blocks_q;
  const int qoff_y = SUB ? NBR * (quarter / g.nq) : 0, qoff_x = SUB ? NBR * (quarter % g.nq) : 0;
  const int n0 = g.batch * g.gh[0] * g.gw[0];
  const int dir = blk >= n0 ? 1 : 0;
  blk -= dir ? n0 : 0;
  const int gw = g.gw[dir], gh = g.gh[dir];
  const int bx = blk % gw, by = (blk / gw) % gh, n = blk / (gw * gh);
  const int tid = threadIdx.x;
  // staged displacement indices: tj in [tj0, tj0 + W), ti in [ti0, ti0 + W)
  const int W = SUB ? WIN : g.D, W2 = W * W, ldg = W2 + 1;
  const int tj0 = !SUB ? 0 : (dir ? 2 * g.R - 4 * qoff_y - 4 * NBR + 1 : 4 * qoff_y - 3);
  const int ti0 = !SUB ? 0 : (dir ? 2 * g.R - 4 * qoff_x - 4 * NBR + 1 : 4 * qoff_x - 3);
  const unsigned w_magic = SUB ? 65536u / (unsigned)WIN + 1u : g.d_magic, w2_magic = SUB ? 0xffffffffu / (unsigned)WIN2 + 1u : g.d2_magic;
  const float* go = g.gout + (long)n * g.g_sb;
  const unsigned sc32 = (unsigned)g.g_sc, sp32 = (unsigned)g.g_sp;
  const int y0 = g.lo_y[dir] + 4 * by, x0 = g.lo_x[dir] + 4 * bx;      // the block's first target pixel, output coordinates
  auto in_img = [&](int y, int x) { return y + g.origin >= 0 && y + g.origin < g.H && x + g.origin >= 0 && x + g.origin < g.W; };
  // ---- stage: every (target pixel, displacement) pair of the block once, coalesced along the layout's contiguous axis --
  // planes (g_sp == 1): the four pixels of a block row are 16 contiguous bytes of a plane; rows (g_sc == 1): a pixel's
  // displacements are one contiguous run.  Validity is folded in here.
  const bool d_fastest = g.g_sc == 1;
  float v[NIT];
  int idx[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + i * 256;
    int m, w;
    if (d_fastest) { m = mdiv32b(e, w2_magic); w = e - m * W2; } else { w = e >> 4; m = e & 15; }
    const bool live = m < 16 && w < W2;
    const int ty = y0 + (m >> 2), tx = x0 + (m & 3);
    const int wy = (int)(((unsigned)min(w, 4095) * w_magic) >> 16);                // (host-made multipliers: no division)
    const int tj = tj0 + wy, ti = ti0 + w - wy * W;
    const int dy = tj - g.R, dx = ti - g.R;
    const int py = dir ? ty - dy : ty, px = dir ? tx - dx : tx;       // p: output pixel;  q = p + d: displaced pixel
    const bool ok = live && tj >= 0 && tj < g.D && ti >= 0 && ti < g.D && py >= 0 && py < g.oh && px >= 0 && px < g.ow &&
                    in_img(py, px) && in_img(py + dy, px + dx);
    // (32-bit offsets: one image's gradOut stays below 2^31 floats -- host-checked)
    const unsigned off = ok ? (unsigned)(tj * g.D + ti) * sc32 + (unsigned)(py * g.ow + px) * sp32 : 0u;
    const float x = (g.ablate & 32) ? 1.f : go[off];
    v[i] = ok ? x : 0.f;
    idx[i] = live ? m * ldg + w : -1;
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i)
    if (idx[i] >= 0) G[idx[i]] = v[i];
  __syncthreads();
  // ---- the band words in register order: thread = (lane, window blocks j, j + 4, ...), the four steps of a block as one float4
  const int lane = tid & 63, j = tid >> 6;
  const int m = lane & 15, k = lane >> 4, tyi = m >> 2, txi = m & 3;
  f32x4* out = reinterpret_cast<f32x4*>(g.band + (long)quarter * g.band_q + g.dir_off[dir]) + (((long)n * gh + by) * gw + bx) * NB2 * 64 + lane;
  const float* Gm = G + m * ldg;
#pragma unroll
  for (int i = 0; i < (NB2 + 3) / 4; ++i) {
    const int e = j + 4 * i;
    if (e >= NB2) break;
    const int qi = e / NBR, qj = e - qi * NBR;
    const int hx = 4 * (qj + qoff_x) + k - txi;                     // halo pixel - target pixel + R
    const int ti = dir ? 2 * g.R - hx : hx;                         // displacement index of the pair
    f32x4 bw;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int hy = 4 * (qi + qoff_y) + t - tyi;
      const int tj = dir ? 2 * g.R - hy : hy;
      const bool in = tj >= 0 && tj < g.D && ti >= 0 && ti < g.D;
      bw[t] = in ? Gm[(tj - tj0) * W + (ti - ti0)] : 0.f;
    }
    if (!(g.ablate & 64) || bw[0] == 12345.f) out[e * 64] = bw;
  }
}

// ------------------------------------------------------------------------------------------------ host side: the plan
struct BPlan {
  int nseg, tiles_per_image, tiles_total, chunk, rem, nchunks, ngroups, items, ahead, use_table;   // chunk, rem: as BGeom
  size_t lds_bytes;
  BSeg seg[kMaxSeg];
  std::vector<unsigned short> table;
};

// ring slots a tile shape can afford: th + ahead, ahead = 4 .. 1 positions beyond the th being read (2 th slots or more keep a
// tall tile from draining the ring at group boundaries)
bool tile_lds(int th, int tw, int nbr, int* ahead, size_t* bytes) {
  const size_t slot = (size_t)4 * 4 * (tw + nbr - 1) * kGC * 4;
  for (int a = 4; a >= 1; --a)
    if ((th + a) * slot <= (size_t)kLdsMax) { *ahead = a; *bytes = (th + a) * slot; return true; }
  return false;
}

// Tiles over GH x GW target blocks: 2 x 4 in the body; the odd columns at the right edge as 4 x w / 2 x w strips (a 4 x 1 tile
// keeps four waves busy where a 2 x 1 tile has two), an odd last row as 1 x 4 / 1 x w tiles.
int bw_segments(int GH, int GW, BSeg* seg) {
  int ns = 0, t0 = 0;
  auto add = [&](int by0, int bx0, int nty, int ntx, int th, int tw) {
    if (nty > 0 && ntx > 0 && ns < kMaxSeg) { seg[ns++] = BSeg{t0, by0, bx0, nty, ntx, th, tw, 1}; t0 += nty * ntx; }
  };
  const int eh = GH / 2 * 2, ew = GW / 4 * 4, rw = GW - ew;
  add(0, 0, eh / 2, ew / 4, 2, 4);
  if (rw == 3) add(0, ew, eh / 2, 1, 2, 3);
  if (rw == 1 || rw == 2) {
    add(0, ew, eh / 4, 1, 4, rw);
    if (eh % 4) add(eh / 4 * 4, ew, 1, 1, 2, rw);
  }
  if (GH & 1) {
    add(eh, 0, 1, ew / 4, 1, 4);
    if (rw) add(eh, ew, 1, 1, 1, rw);
  }
  return ns;
}

// makespan of `dur` (sorted descending) dealt greedily to `ncu` machines
double makespan(const std::vector<double>& dur, int ncu) {
  std::vector<double> heap(ncu, 0.0);   // min-heap of machine finish times
  auto cmp = [](double a, double b) { return a > b; };
  for (double d : dur) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    heap.back() += d;
    std::push_heap(heap.begin(), heap.end(), cmp);
  }
  return *std::max_element(heap.begin(), heap.end());
}

// gscale: matrix-pipe time of one channel group relative to NBR x NBR window blocks (window radius > 8: the four quarters of a group)
bool plan_bwd(int batch, int eh, int ew, int nbr, int ngroups, int ncu, double gscale, BPlan* out) {
  const int GH = (eh + 3) / 4, GW = (ew + 3) / 4;
  BPlan p;
  p.nseg = bw_segments(GH, GW, p.seg);
  if (p.nseg == 0 || ncu < 1) return false;
  p.tiles_per_image = 0;
  size_t lds = 0;
  for (int s = 0; s < p.nseg; ++s) {
    p.tiles_per_image += p.seg[s].nty * p.seg[s].ntx;
    int a; size_t b;
    if (!tile_lds(p.seg[s].th, p.seg[s].tw, nbr, &a, &b)) return false;
    p.seg[s].ahead = a;
    lds = std::max(lds, b);
  }
  p.ahead = 0; p.lds_bytes = lds;
  p.tiles_total = p.tiles_per_image * batch;
  p.ngroups = ngroups;
  // cost of one (tile, group) unit in matrix-pipe time: the waves of a tile sit ceil(waves / 4) deep on a SIMD; a workgroup pays
  // a set-up (ring fill, band fetch, the drain of the workgroup before it) worth about three quarters of a two-deep unit
  std::vector<double> tile_cost;
  for (int s = 0; s < p.nseg; ++s)
    for (int i = 0; i < p.seg[s].nty * p.seg[s].ntx; ++i) tile_cost.push_back(gscale * ((p.seg[s].th * p.seg[s].tw + 3) / 4));
  // (fitted to forced-chunk sweeps on MI355X, measured time / simulated makespan within 2 - 6 %: conv5 / conv4 at radius 8 price a work
  //  item's set-up at 1.5 units; the quarter-walking kernel's steps are slower per unit and its sweeps rank as if the set-up were free)
  const double setup = gscale > 1.0 ? 0.3 : 1.5;
  double best = 1e30;
  int best_n = 1;
  std::vector<double> dur;
  for (int nchunks = 1; nchunks <= ngroups; ++nchunks) {
    if ((long)nchunks * p.tiles_total > 65535) break;                          // (16-bit item numbers; more items than that never pay)
    if ((long)nchunks * p.tiles_total > kTable && best < 1e30) break;          // (... nor does losing the dispatch order that rides in the arguments)
    const int base = ngroups / nchunks, rem = ngroups % nchunks;               // the groups are dealt evenly: chunk lengths differ by one at most
    dur.clear();
    for (int ci = 0; ci < nchunks; ++ci) {
      const int len = base + (ci < rem ? 1 : 0);
      for (int b = 0; b < batch; ++b)
        for (double c : tile_cost) dur.push_back(setup + len * c);
    }
    std::sort(dur.begin(), dur.end(), [](double a, double b) { return a > b; });
    const double ms = makespan(dur, ncu);
    if (ms < best - 1e-9) { best = ms; best_n = nchunks; }
  }
  p.nchunks = best_n;
  if (const char* e = getenv("DTT_CORR_BWD_CHUNK")) {   // developer / test switch: force the channel groups per work item (at most)
    const int c = atoi(e);
    if (c >= 1) p.nchunks = (ngroups + std::min(c, ngroups) - 1) / std::min(c, ngroups);
    while ((long)p.nchunks * p.tiles_total > 65535) --p.nchunks;
  }
  const int nchunks = p.nchunks;
  p.chunk = ngroups / nchunks; p.rem = ngroups % nchunks;
  p.items = nchunks * p.tiles_total;
  // ---- dispatch order.  Block b runs on XCD b % 8 and the blocks of an XCD start in order: every XCD's queue gets its share
  // of each duration class, longest first; within a class an XCD's share is a contiguous run of (chunk, tile) = neighbouring
  // tiles working on the same channel groups.
  p.use_table = p.items <= kTable ? 1 : 0;
  if (p.use_table) {
    struct It { double dur; int id; };
    std::vector<It> its(p.items);
    for (int ci = 0; ci < nchunks; ++ci) {
      const int len = p.chunk + (ci < p.rem ? 1 : 0);
      for (int t = 0; t < p.tiles_total; ++t) its[ci * p.tiles_total + t] = It{setup + len * tile_cost[t % p.tiles_per_image], ci * p.tiles_total + t};
    }
    std::stable_sort(its.begin(), its.end(), [](const It& a, const It& b) { return a.dur > b.dur; });
    std::vector<std::vector<int>> q(8);
    // (measured on MI355X, profiles/r06_corr_bwd_order.txt: tile-major queues win where every chunk has the same length -- conv5 at
    //  radius 8: 195 -> 175 us for both gradients -- and for the quarter-walking kernel, whose band words are re-read per group and
    //  quarter: d = 16 conv5 265 -> 254 us with a quarter of the fetch traffic; with uneven chunks (conv4 at radius 8) they break the
    //  longest-first order and lose 6 %)
    const int gang = getenv("DTT_CORR_BWD_GANG") ? atoi(getenv("DTT_CORR_BWD_GANG")) : (gscale > 1.0 || p.rem == 0) ? 4 : 0;   // env: developer A/B switch
    size_t lo0 = 0;
    if (gang > 0) {
      // TILE-major queues for the costliest tile class (the 2 x 4 body tiles): an XCD owns a run of neighbouring tiles and walks them
      // `gang` tiles at a time through all channel chunks -- the work items that are resident together share their tiles' band words
      // in that XCD's L2 (and neighbouring tiles of a gang part of their halos).  The smaller classes are dealt item by item below.
      std::vector<int> tl;
      double cmax = 0;
      for (double c : tile_cost) cmax = std::max(cmax, c);
      for (int t = 0; t < p.tiles_total; ++t)
        if (tile_cost[t % p.tiles_per_image] == cmax) tl.push_back(t);
      const size_t n = tl.size();
      for (int x = 0; x < 8; ++x) {
        const size_t a = n * x / 8, b = n * (x + 1) / 8;
        for (size_t g0 = a; g0 < b; g0 += gang)
          for (int ci = 0; ci < nchunks; ++ci)
            for (size_t i = g0; i < std::min(b, g0 + gang); ++i) q[x].push_back(ci * p.tiles_total + tl[i]);
      }
      // (the items of that class lead `its`, whatever their chunk length: skip them)
      std::vector<It> rest;
      for (const It& e : its)
        if (tile_cost[(e.id % p.tiles_total) % p.tiles_per_image] != cmax) rest.push_back(e);
      its.swap(rest);
    }
    for (size_t lo = lo0; lo < its.size();) {
      size_t hi = lo;
      while (hi < its.size() && its[hi].dur == its[lo].dur) ++hi;
      const size_t n = hi - lo;
      for (int x = 0; x < 8; ++x)
        for (size_t i = lo + n * x / 8; i < lo + n * (x + 1) / 8; ++i) q[x].push_back(its[i].id);
      lo = hi;
    }
    // the hardware's queue lengths: XCD x runs blocks x, x + 8, ... -> items / 8 (+ 1 for x < items % 8)
    for (;;) {
      int lng = -1, sht = -1;
      for (int x = 0; x < 8; ++x) {
        const int want = p.items / 8 + (x < p.items % 8 ? 1 : 0);
        if ((int)q[x].size() > want && lng < 0) lng = x;
        if ((int)q[x].size() < want && sht < 0) sht = x;
      }
      if (lng < 0 || sht < 0) break;
      q[sht].push_back(q[lng].back());
      q[lng].pop_back();
    }
    p.table.assign(p.items, 0);
    for (int x = 0; x < 8; ++x)
      for (size_t k = 0; k < q[x].size(); ++k) p.table[x + 8 * k] = (unsigned short)q[x][k];
  }
  *out = p;
  return true;
}

struct PlanKey { int batch, eh, ew, nbr, ngroups, ncu, forced_chunk, gscale_pct; };
bool operator==(const PlanKey& a, const PlanKey& b) { return memcmp(&a, &b, sizeof(PlanKey)) == 0; }

const BPlan* cached_plan(const PlanKey& k) {
  static std::mutex mu;
  static std::vector<std::pair<PlanKey, BPlan*>> cache;   // a handful of shapes per process; entries live as long as the process
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : cache)
    if (e.first == k) return e.second;
  BPlan* p = new BPlan();
  if (!plan_bwd(k.batch, k.eh, k.ew, k.nbr, k.ngroups, k.ncu, k.gscale_pct / 100.0, p)) { delete p; p = nullptr; }
  cache.emplace_back(k, p);
  return p;
}

template <int NBR, bool MULTI = false>
int launch_stream(const BGeom& g, int items, size_t lds_bytes, hipStream_t stream) {
  static DttDeviceOnce once;
  bool& done = once.here();
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_bwd_stream_kernel<NBR, MULTI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
    DTT_REQUIRE(e == hipSuccess, "correlation backward (streamed): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    done = true;
  }
  hipLaunchKernelGGL((corr_bwd_stream_kernel<NBR, MULTI>), dim3(items), dim3(kThreads), lds_bytes, stream, g);
  DTT_CHECK_LAUNCH("corr_bwd_stream_kernel");
  return 1;
}

// target ranges of the two directions in output coordinates (inclusive); false: empty
bool target_range(bool wrt2, int oh, int ow, int H, int W, int origin, int R, int* lo, int* hi_y, int* hi_x) {
  if (!wrt2) {   // output pixels whose lattice pixel lies inside the image
    *lo = std::max(0, -origin);
    *hi_y = std::min(oh - 1, H - 1 - origin); *hi_x = std::min(ow - 1, W - 1 - origin);
  } else {       // displaced pixels q = p + d that fall inside the image
    *lo = std::max(-origin, -R);
    *hi_y = std::min(H - 1 - origin, oh - 1 + R); *hi_x = std::min(W - 1 - origin, ow - 1 + R);
  }
  return *hi_y >= *lo && *hi_x >= *lo;
}

}  // namespace

// 1 if the band-stationary streamed kernels cover this geometry (else the caller takes the round-1 kernels)
extern "C" int dtt_correlation_backward_stream_supported(int ic, int kernel_size, int max_displacement, int stride1, int stride2) {
  if (kernel_size != 1 || stride1 != stride2 || stride1 < 1 || max_displacement % stride1 != 0) return 0;
  const int R = max_displacement / stride1;
  return R >= 1 && R <= 16 && ic % kGC == 0;   // (radius 9 .. 16: the window in four quarters of 5 x 5 blocks)
}

// bytes of workspace dtt_correlation_backward_nhwc_strided needs for this geometry (the band words of both directions); 0 where the
// streamed kernels do not apply
extern "C" size_t dtt_correlation_backward_workspace_bytes(int batch, int ic, int ih, int iw, int pad_size, int kernel_size,
                                                           int max_displacement, int stride1, int stride2) {
  if (!dtt_correlation_backward_stream_supported(ic, kernel_size, max_displacement, stride1, stride2) || batch < 1) return 0;
  if ((max_displacement - pad_size) % stride1 != 0) return 0;
  int oc, oh, ow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow)) return 0;
  const int s = stride1, R = max_displacement / s, nbr = R <= 4 ? 3 : 5;
  const int H = (ih + s - 1) / s, W = (iw + s - 1) / s, origin = (max_displacement - pad_size) / s;
  const int nbr_full = 1 + (R + 1) / 2, nq = (nbr_full + nbr - 1) / nbr;   // window radius > 8: nq x nq quarters, all resident (one launch walks them)
  size_t total = 0;
  for (int dir = 0; dir < 2; ++dir) {
    int lo, hy, hx;
    if (!target_range(dir == 1, oh, ow, H, W, origin, R, &lo, &hy, &hx)) continue;
    total += (size_t)batch * ((hy - lo + 4) / 4) * ((hx - lo + 4) / 4) * nbr * nbr * 4 * 64 * sizeof(float);
  }
  return total * nq * nq;
}

// Both gradients, channels-last inputs and gradients; gradOut[n, d, p] at gradOutput[n * g_sb + d * g_sc + p * g_sp].
// which: 1 = gradInput1 only, 2 = gradInput2 only, 3 = both.  phase: 1 = lay out the band words in the workspace (reads gradOutput only),
// 2 = the gradients from a workspace that phase 1 filled (reads the maps and the workspace), 3 = both, one after the other.
int dtt_corr_bwd_stream(const float* gradOutput, long g_sb, long g_sc, long g_sp, int gob, int goh, int gow, const float* input1, int ic,
                        int ih, int iw, const float* input2, float* gradInput1, float* gradInput2, int pad_size, int max_displacement,
                        int stride, int which, int phase, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const int s = stride, R = max_displacement / s;
  DTT_REQUIRE((max_displacement - pad_size) % s == 0, "correlation backward (channels-last): displacement - padding must be a multiple of the stride");
  DTT_REQUIRE((long)ih * iw * ic * 4 < 0xffffffffl, "correlation backward (channels-last): one image exceeds the 32-bit offset range");
  {
    const long D2l = (long)(2 * R + 1) * (2 * R + 1), npx = (long)goh * gow;
    DTT_REQUIRE(g_sc >= 0 && g_sp >= 0 && (D2l - 1) * g_sc + (npx - 1) * g_sp < (1l << 31),
                "correlation backward (channels-last): gradOutput strides exceed the 32-bit offset range of one image");
  }
  const size_t ws_need = dtt_correlation_backward_workspace_bytes(gob, ic, ih, iw, pad_size, 1, max_displacement, s, s);
  DTT_REQUIRE(workspace && workspace_bytes >= ws_need && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0,
              "correlation backward (channels-last): workspace of %zu bytes needed (dtt_correlation_backward_workspace_bytes), got %zu",
              ws_need, workspace ? workspace_bytes : (size_t)0);
  BGeom g;
  memset(&g, 0, sizeof(g));
  g.C = ic;
  g.H = (ih + s - 1) / s; g.W = (iw + s - 1) / s;
  g.sb = (long)ih * iw * ic;
  g.sx4 = (unsigned)((long)s * ic * 4); g.sy4 = (unsigned)((long)s * iw * ic * 4);
  g.stride = s; g.ih = ih; g.iw = iw; g.px4 = (unsigned)((long)ic * 4); g.py4 = (unsigned)((long)iw * ic * 4);
  g.oh = goh; g.ow = gow; g.origin = (max_displacement - pad_size) / s;
  g.R = R; g.D = 2 * R + 1;
  g.inv = 1.f / (float)ic;
  static const int ablate = getenv("DTT_CORR_BWD_ABLATE") ? atoi(getenv("DTT_CORR_BWD_ABLATE")) : 0;
  g.ablate = ablate;
  const int nbr = R <= 4 ? 3 : 5;
  const int ncu = dtt_device_cus();

  // ---- the band words of both directions (and, window radius > 8, of all four window quarters): one small launch
  const int nbr_full = 1 + (R + 1) / 2, nq = (nbr_full + nbr - 1) / nbr;
  DTT_REQUIRE(nq == 1 || (nq == 2 && nbr == 5), "correlation backward (streamed): window radius %d not covered", R);
  BandGeom bg;
  memset(&bg, 0, sizeof(bg));
  bg.gout = gradOutput; bg.g_sb = g_sb; bg.g_sc = g_sc; bg.g_sp = g_sp;
  bg.band = static_cast<float*>(workspace);
  bg.oh = goh; bg.ow = gow; bg.origin = g.origin; bg.H = g.H; bg.W = g.W; bg.R = R; bg.D = g.D; bg.batch = gob;
  bg.d_magic = 65536u / (unsigned)g.D + 1u; bg.d2_magic = 0xffffffffu / (unsigned)(g.D * g.D) + 1u;
  bg.ablate = ablate;
  int lo[2], hy[2], hx[2];
  bool live[2];
  long off = 0;
  int band_blocks = 0;
  for (int dir = 0; dir < 2; ++dir) {
    live[dir] = target_range(dir == 1, goh, gow, g.H, g.W, g.origin, R, &lo[dir], &hy[dir], &hx[dir]);
    bg.lo_y[dir] = bg.lo_x[dir] = live[dir] ? lo[dir] : 0;
    bg.gh[dir] = live[dir] ? (hy[dir] - lo[dir] + 4) / 4 : 0;
    bg.gw[dir] = live[dir] ? (hx[dir] - lo[dir] + 4) / 4 : 0;
    bg.dir_off[dir] = off;
    off += (long)gob * bg.gh[dir] * bg.gw[dir] * nbr * nbr * 4 * 64;
    band_blocks += gob * bg.gh[dir] * bg.gw[dir];
  }
  bg.nq = nq; bg.blocks_q = band_blocks; bg.band_q = off;
  const size_t bytes = (size_t)gob * ic * ih * iw * sizeof(float);
  // pixels that no lattice point maps to (stride > 1, pad != displacement) keep a zero gradient; on the dense lattice of conv4 /
  // conv5 (stride 1, pad == displacement) both kernels write every element themselves
  const bool dense = s == 1 && g.origin == 0 && goh == g.H && gow == g.W;
  // a strided lattice that the outputs cover completely (conv3: stride 2, pad == displacement): every lattice pixel is a target of
  // both directions and zeroes the stride x stride cell of image pixels behind it -- again every element is written here
  const bool lattice_dense = s > 1 && s <= 4 && g.origin == 0 && goh == g.H && gow == g.W;
  for (int dir = 0; dir < 2; ++dir)
    if ((phase & 2) && (which & (dir ? 2 : 1)) && !dense && !lattice_dense)
      DTT_REQUIRE(hipMemsetAsync(dir ? gradInput2 : gradInput1, 0, bytes, stream) == hipSuccess, "correlation backward: memset failed");
  if (band_blocks > 0 && (phase & 1)) {
    if (nbr == 3) hipLaunchKernelGGL((corr_bwd_band_kernel<3, false>), dim3(band_blocks), dim3(256), 0, stream, bg);
    else if (nq == 1) hipLaunchKernelGGL((corr_bwd_band_kernel<5, false>), dim3(band_blocks), dim3(256), 0, stream, bg);
    else hipLaunchKernelGGL((corr_bwd_band_kernel<5, true>), dim3(band_blocks * nq * nq), dim3(256), 0, stream, bg);
    DTT_CHECK_LAUNCH("corr_bwd_band_kernel");
  }
  if (!(phase & 2)) return 1;
  // Window radius > 8 (BASELINE configs[4]: d = 16, 33 x 33 displacements = 9 x 9 window blocks, 324 band words per lane): the window
  // is covered in quarters of NBR x NBR = 5 x 5 blocks INSIDE the launch -- per channel group the four quarters' halos stream through
  // the ring one after the other into the same accumulators, the band registers follow (corr_bwd_stream_kernel<5, true>): every gradient
  // element is still written exactly once, no read-modify-write, in a fixed order.
  g.nyl = nq > 1 ? nbr_full - nbr : nbr; g.nxl = g.nyl;
  g.band_q = off;
  g.cell_fill = lattice_dense ? 1 : 0;
  g.nimg = gob;
  // matrix-pipe time of a channel group in units of NBR x NBR window blocks (what the plan's set-up cost is priced against)
  const int gscale_pct = nq > 1 ? (int)(100.0 * nbr_full * nbr_full / (nbr * nbr)) : 100;
  static const bool no_merge = getenv("DTT_CORR_BWD_NO_MERGE") != nullptr;      // developer A/B switch: one launch per direction
  // both directions as ONE grid where their target ranges coincide (pad == displacement: every lattice pixel is a target of both):
  // the plan sees 2 x batch images, image n >= batch is direction 1's image n - batch -- one dispatch tail instead of two
  const bool merged = which == 3 && live[0] && live[1] && lo[0] == lo[1] && hy[0] == hy[1] && hx[0] == hx[1] && !no_merge;
  for (int dir = 0; dir < 2; ++dir) {
    const bool wrt2 = dir == 1;
    if (merged && dir == 1) break;
    if (!merged && (!(which & (wrt2 ? 2 : 1)) || !live[dir])) continue;
    g.lo_y = g.lo_x = lo[dir]; g.hi_y = hy[dir]; g.hi_x = hx[dir];
    g.gh = bg.gh[dir]; g.gw = bg.gw[dir];
    g.band = bg.band + bg.dir_off[dir];
    const PlanKey key{merged ? 2 * gob : gob, g.hi_y - g.lo_y + 1, g.hi_x - g.lo_x + 1, nbr, ic / kGC, ncu,
                      getenv("DTT_CORR_BWD_CHUNK") ? atoi(getenv("DTT_CORR_BWD_CHUNK")) : 0, gscale_pct};
    const BPlan* p = cached_plan(key);
    DTT_REQUIRE(p != nullptr, "correlation backward (streamed): no plan for %d x %d targets, radius %d", key.eh, key.ew, R);
    g.other[0] = wrt2 ? input1 : input2; g.grad[0] = wrt2 ? gradInput2 : gradInput1;
    g.other[1] = input1; g.grad[1] = gradInput2;            // (read by a merged launch only)
    g.nseg = p->nseg; g.tiles_per_image = p->tiles_per_image; g.tiles_total = p->tiles_total;
    for (int i = 0; i < kMaxSeg; ++i) g.seg[i] = i < p->nseg ? p->seg[i] : BSeg{0x7fffffff, 0, 0, 1, 1, 1, 1, 1};
    g.chunk = p->chunk; g.rem = p->rem; g.ngroups = p->ngroups; g.use_table = p->use_table;
    if (p->use_table) memcpy(g.table, p->table.data(), sizeof(unsigned short) * p->items);
    // (one kernel for both directions: which gradient a work item computes is a matter of its band words and of `other`)
    const int ok = nbr == 3 ? launch_stream<3>(g, p->items, p->lds_bytes, stream)
                            : nq > 1 ? launch_stream<5, true>(g, p->items, p->lds_bytes, stream) : launch_stream<5>(g, p->items, p->lds_bytes, stream);
    if (!ok) return 0;
  }
  return 1;
}

// Test hook (pure host code): replays every work item of the plan the launcher would use through the kernel's own item decode and
// counts, per (image, target block, channel group), how many waves own it.  Returns 1 iff each is owned exactly once, the dispatch
// table is a permutation of the items, and every tile shape fits its ring.
extern "C" int dtt_correlation_backward_plan_check(int batch, int target_h, int target_w, int window_radius, int channels,
                                                   int compute_units) {
  if (window_radius < 1 || window_radius > 16 || channels % kGC != 0 || batch < 1 || target_h < 1 || target_w < 1) return 0;
  const int R = window_radius, nbr = R <= 4 ? 3 : 5;
  const int nbr_full = 1 + (R + 1) / 2;
  BPlan p;
  if (!plan_bwd(batch, target_h, target_w, nbr, channels / kGC, compute_units > 0 ? compute_units : 256,
                R > 8 ? (int)(100.0 * nbr_full * nbr_full / (nbr * nbr)) / 100.0 : 1.0, &p)) return 0;
  BGeom g;
  memset(&g, 0, sizeof(g));
  g.nseg = p.nseg; g.tiles_per_image = p.tiles_per_image; g.tiles_total = p.tiles_total;
  for (int i = 0; i < kMaxSeg; ++i) g.seg[i] = i < p.nseg ? p.seg[i] : BSeg{0x7fffffff, 0, 0, 1, 1, 1, 1, 1};
  g.chunk = p.chunk; g.rem = p.rem; g.ngroups = p.ngroups;
  const int GH = (target_h + 3) / 4, GW = (target_w + 3) / 4, NG = channels / kGC;
  std::vector<unsigned char> owned((size_t)batch * GH * GW * NG, 0);
  std::vector<unsigned char> seen(p.items, 0);
  if (p.lds_bytes > (size_t)kLdsMax) return 0;
  for (int b = 0; b < p.items; ++b) {
    const int item = p.use_table ? (int)p.table[b] : b;
    if (item < 0 || item >= p.items || seen[item]) return 0;
    seen[item] = 1;
    const BItem it = bw_decode(g, item);
    if (it.n < 0 || it.n >= batch || it.ng < 1 || it.g0 + it.ng > NG || it.th * it.tw > kComp) return 0;
    if (4 * (it.tw + nbr - 1) > kMaxNI * kLoad) return 0;
    if ((size_t)(it.th + it.ahead) * 4 * 4 * (it.tw + nbr - 1) * kGC * 4 > p.lds_bytes || it.ahead < 1) return 0;
    for (int w = 0; w < it.th * it.tw; ++w) {
      const int by = it.Y0 / 4 + w / it.tw, bx = it.X0 / 4 + w % it.tw;
      if (by >= GH || bx >= GW) return 0;
      for (int gi = it.g0; gi < it.g0 + it.ng; ++gi) {
        unsigned char& c = owned[(((size_t)it.n * GH + by) * GW + bx) * NG + gi];
        if (c) return 0;
        c = 1;
      }
    }
  }
  for (unsigned char c : owned)
    if (!c) return 0;
  return 1;
}

// developer / test hook: the plan's shape (work items, channel groups per item, LDS bytes, whether the order rides in the arguments)
extern "C" int dtt_correlation_backward_plan(int batch, int target_h, int target_w, int window_radius, int channels, int compute_units,
                                             int* items, int* chunk, int* lds_bytes, int* table) {
  if (window_radius < 1 || window_radius > 16 || channels % kGC != 0) return 0;
  const int R = window_radius, nbr = R <= 4 ? 3 : 5;
  const int nbr_full = 1 + (R + 1) / 2;
  BPlan p;
  if (!plan_bwd(batch, target_h, target_w, nbr, channels / kGC, compute_units > 0 ? compute_units : 256,
                R > 8 ? (int)(100.0 * nbr_full * nbr_full / (nbr * nbr)) / 100.0 : 1.0, &p)) return 0;
  if (items) *items = p.items;
  if (chunk) *chunk = p.chunk + (p.rem ? 1 : 0);   // (the longest work item's channel groups)
  if (lds_bytes) *lds_bytes = (int)p.lds_bytes;
  if (table) *table = p.use_table;
  return 1;
}

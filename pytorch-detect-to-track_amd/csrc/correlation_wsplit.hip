// Cross-frame correlation on channels-last maps, WINDOW-SPLIT form (gfx950): the forward of Correlation_forward
// (correlation/src/correlation_cuda_kernel.cu:34-106) for kernel_size 1, stride1 == stride2, max_displacement / stride <= 16
// -- the three correlations of D&T (rfcn.py:58-60, 170-172) and BASELINE configs[4]'s d = 16 -- as ONE launch with no
// partial sums anywhere.
//
//   out[p, q] = 1/C * sum_c f1[p, c] * f2[q, c],  |q - p| <= R  is a banded matrix product.  A wave owns one 4 x 4 block of
//   frame-t pixels (16 MFMA columns) and multiplies it against 4 x 4 blocks of frame-(t+tau) pixels (16 MFMA rows) with the
//   exact-f32 v_mfma_f32_16x16x4_f32 (an fma chain, bit for bit): NBR x NBR such blocks cover its window, NBR = 1 + ceil(R/2).
//   At batch 2 a 38 x 67 map has only 340 pixel blocks for 1024 SIMDs.  Round 2 filled the chip by cutting the CHANNELS of a
//   tile into slices that met through slabs in memory (73.7 MB out and back per conv5 op and a 38 us last-arriver tail,
//   measured).  Here the WINDOW is cut instead: the NBR^2 window blocks of a pixel block are dealt to `parts` waves in
//   row-major runs (25 -> 8 + 8 + 9), every wave runs ALL channels for its run, and what it accumulates is final:
//   no slabs, no tickets, no reducer, no workspace, nothing to race on, bit-identical from run to run by construction.
//
//   * Workgroup = 8 compute waves (two per SIMD: a wave PAIR shares a pixel block's window run and takes alternate channel
//     chunks) + 4 loader waves, one tile of up to four pixel blocks (2 x 2, or 4 x 1 / 1 x 4 / smaller along odd map edges)
//     x one run of window parts.  Work items are laid out by a host-side plan
//     (segments of identical tiles) so that tiles x parts fills the chip: 2 x (42 x 3 + 2) = 256 workgroups at 600 px, or
//     five parts (426 short workgroups, two rounds) when the caller wants CUs left free (`max_workgroups`).
//   * Loaders stage, per 16-channel chunk, the tile's pixels of frame t and only the halo ROWS its window run touches
//     (12 - 16 of 24 at three parts) global -> LDS by LDS-DMA in the scalar-base form (SALU + VMEM only), into a ring of
//     3 - 8 slots; they run up to nslot - 1 chunks ahead with counted s_waitcnt vmcnt.  The 16-byte pieces of a pixel's
//     64-byte chunk row are XOR-swizzled on the source side so that every ds_read_b128 is bank-conflict free.
//   * Compute waves: one barrier per chunk; a wave reads the operands of ITS chunk (one ds_read_b128 per window block = four
//     MFMA k-steps) in two bursts in two consecutive barrier intervals and runs the MFMAs of a burst one interval later, so a
//     read burst always sits underneath the partner wave's MFMAs; the two waves' sums meet in LDS at the end (fixed order).
//   * Epilogue: each workgroup assembles ITS window entries in LDS (1/C applied, zero where p or q lies in the padding) and
//     streams them out in the caller's layout; the three parts of a tile write disjoint entries of the same rows.
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <vector>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kKC = 16;                  // channels per ring slot
constexpr int kPPX = 64;                 // frame-t pixels per slot: up to four 4 x 4 blocks, block-major
constexpr int kNComp = 8, kNLoad = 4, kThreads = (kNComp + kNLoad) * 64;   // 2 compute waves + 1 loader per SIMD
constexpr int kMaxNI = 12;               // DMA instructions per loader per chunk (4 loaders x 12 x 1 KB = 48 KB per slot)
constexpr int kMaxSeg = 6;
constexpr int kLdsMax = 144 * 1024;       // 16 KB of the CU's 160 stay free: small kernels of other streams (the NMS mask kernel's
                                          // one-wave workgroups with 1 KB of LDS) can still become resident beside a correlation workgroup
constexpr double kDmaBytesPerClk = 32.0;   // planning figure: LDS-DMA into one CU beside a running MFMA stream

struct WSeg { int item0, nitems, by0, bx0, nty, ntx, th, tw, wpt; unsigned wpt_magic, tiles_magic, nty_magic; };   // magics: 2^32 / d + 1

struct WGeom {
  const float* f1; const float* f2;      // frame t / t+tau, channels-last
  long sy, sx, sb;                       // floats between vertically / horizontally adjacent lattice pixels, between images
  unsigned sy4, sx4;                     // the same in bytes (one image stays below 4 GB: the DMA offsets are 32-bit)
  int C, H, W;                           // channels, lattice size
  int oh, ow, origin;                    // output size; output (y, x) <-> lattice pixel (origin + y, origin + x)
  int R, D, nbr, nblk, parts;
  unsigned nbr_magic, d_magic;           // x / nbr == (x * nbr_magic) >> 16 for x < 4096; e / D likewise (host-checked ranges)
  unsigned parts_magic;                  // 2^32 / parts + 1
  float* out; long out_sb, out_sc, out_sp;   // element (n, d, y, x) at out[n*sb + d*sc + (y*ow + x)*sp]
  int nseg; WSeg seg[kMaxSeg];
  int ring_bytes;                        // LDS given to the ring: every workgroup cuts it into slots of ITS halo size (at most 8)
  int ablate;                            // developer timing experiments (DTT_CORR_WS_ABLATE): 1 no DMA, 2 no MFMA, 4 no epilogue
};

__device__ __forceinline__ void dma16w(const char* sbase, unsigned voff, unsigned lds_addr) {
  // m0 (the DMA's LDS base) is put back inside the statement: it is a reserved register the compiler neither allocates nor saves
  // around inline asm -- a statement that merely listed it as clobbered would rely on hipcc never keeping a value of its own there
  unsigned keep_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep_m0)
               : "s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory");
}
__device__ __forceinline__ const char* uptrw(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  return (const char*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
// piece swizzle: the 16-byte piece g of the 64-byte chunk row of a pixel in row (key % 4) of its 4 x 4 block sits in slot
// g ^ kSwz[key].  With pixel index % 4 == column-in-block this makes the four 16-lane groups of a ds_read_b128
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md, LDS table) hit 16 different 16-byte bank quads each.
__device__ __forceinline__ int swz(int key) { return (0x1320 >> ((key & 3) * 4)) & 3; }   // {0, 2, 3, 1}

// s_waitcnt vmcnt(n) with a wave-uniform runtime n (the immediate must be a constant)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define DTT_W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define DTT_W8(b) DTT_W1(b) DTT_W1(b + 1) DTT_W1(b + 2) DTT_W1(b + 3) DTT_W1(b + 4) DTT_W1(b + 5) DTT_W1(b + 6) DTT_W1(b + 7)
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    DTT_W1(1) DTT_W1(2) DTT_W1(3) DTT_W1(4) DTT_W1(5) DTT_W1(6) DTT_W1(7)
    DTT_W8(8) DTT_W8(16) DTT_W8(24) DTT_W8(32) DTT_W8(40) DTT_W8(48)
    DTT_W1(56) DTT_W1(57) DTT_W1(58) DTT_W1(59) DTT_W1(60) DTT_W1(61) DTT_W1(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef DTT_W8
#undef DTT_W1
}

// n / d for 0 <= n < 65536 with magic = 65536 / d + 1 (exact while n * (magic * d - 65536) < 65536: the host checks the range)
__host__ __device__ __forceinline__ int mdiv(int n, unsigned magic) { return (int)(((unsigned)n * magic) >> 16); }
// n / d with magic = (2^32 - 1) / d + 1 (which wraps to 0 for d == 1): exact while n * d < 2^32 (host-checked)
__host__ __device__ __forceinline__ int mdiv32(int n, unsigned magic) {
  return magic ? (int)(((unsigned long long)(unsigned)n * magic) >> 32) : n;
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// What one workgroup does, worked out from its item number and the plan in the kernel arguments (wave-uniform; the same code
// runs on the host in dtt_correlation_nhwc_plan_check, which replays every item of a plan and counts who computes what).
struct WItem {
  int n;                    // image
  int Y0, X0;               // tile origin, output pixels
  int th, tw, nb, nb_shift, tw_shift;
  int jw;                   // workgroup jw of its tile: wave task t = 4 * jw + wave -> pixel block t % nb, window part t / nb
  int q_lo, q_hi;           // window blocks [q_lo, q_hi) of every pixel block of the tile belong to this workgroup
  int r0, r1;               // the window block rows they lie in
  int hrows, HC;            // halo pixels staged per chunk: hrows x HC
};

__host__ __device__ __forceinline__ int ws_q0(const WGeom& g, int p) { return mdiv32(g.nblk * p, g.parts_magic); }

__host__ __device__ __forceinline__ WItem ws_decode(const WGeom& g, int item) {
  WSeg sg = g.seg[0];
#pragma unroll
  for (int i = 1; i < kMaxSeg; ++i)
    if (i < g.nseg && item >= g.seg[i].item0) sg = g.seg[i];
  WItem w;
  const int local = item - sg.item0;
  const int ntiles = sg.nty * sg.ntx;
  // (divisions by plan constants through host-made multipliers: the set-up sits in front of the first DMA)
  const int wt = mdiv32(local, sg.wpt_magic);
  w.jw = local - wt * sg.wpt;
  w.n = mdiv32(wt, sg.tiles_magic);
  const int tile = wt - w.n * ntiles;
  const int txi = mdiv32(tile, sg.nty_magic), tyi = tile - txi * sg.nty;         // column-major: an XCD's run of items is a vertical strip
  w.th = sg.th; w.tw = sg.tw; w.nb = sg.th * sg.tw;
  w.nb_shift = w.nb == 4 ? 2 : w.nb == 2 ? 1 : 0;                                // nb, tw in {1, 2, 4}
  w.tw_shift = w.tw == 4 ? 2 : w.tw == 2 ? 1 : 0;
  w.Y0 = 4 * (sg.by0 + tyi * sg.th); w.X0 = 4 * (sg.bx0 + txi * sg.tw);
  const int p_lo = (4 * w.jw) >> w.nb_shift, p_hi = min(g.parts - 1, (4 * w.jw + 3) >> w.nb_shift);
  w.q_lo = ws_q0(g, p_lo); w.q_hi = ws_q0(g, p_hi + 1);
  w.r0 = mdiv(w.q_lo, g.nbr_magic); w.r1 = mdiv(w.q_hi - 1, g.nbr_magic);
  w.hrows = 4 * (w.r1 - w.r0 + sg.th); w.HC = 4 * (g.nbr - 1 + sg.tw);
  return w;
}

#ifdef DTT_WS_TRACE   // developer build (tools/build_ws_trace.sh): shader-clock stamps of every workgroup's waves 0 / 4 / 8
__device__ unsigned long long dtt_ws_trace[512 * 3 * 16];
#define WS_STAMP(slot) do { if ((threadIdx.x & 63) == 0 && (wave & 3) == 0 && blockIdx.x < 512) \
    dtt_ws_trace[(blockIdx.x * 3 + (wave >> 2)) * 16 + (slot)] = (slot) == 15 ? wall_clock64() : __builtin_readcyclecounter(); } while (0)
#else
#define WS_STAMP(slot) do {} while (0)
#endif

template <int NACC>
__global__ __launch_bounds__(kThreads) void corr_wsplit_kernel(WGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  WS_STAMP(0);
  WS_STAMP(15);
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);

  // ---- which tile, which run of window parts (all wave-uniform)
  const WItem wi = ws_decode(g, item);
  const int jw = wi.jw, n = wi.n, nb = wi.nb, tw = wi.tw, nb_shift = wi.nb_shift, tw_shift = wi.tw_shift;
  const int Y0 = wi.Y0, X0 = wi.X0, q_lo = wi.q_lo, q_hi = wi.q_hi, r0 = wi.r0, r1 = wi.r1, hrows = wi.hrows, HC = wi.HC;
  auto q0 = [&](int p) { return ws_q0(g, p); };                                  // window blocks [q0(p), q0(p+1)) belong to part p
  const int n_instr = (kPPX + hrows * HC) / 16;
  const int slot_bytes = (kPPX + hrows * HC) * kKC * 4, nslot = min(8, g.ring_bytes / slot_bytes);
  const int nch = g.C / kKC;
  const unsigned lds0 = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)lds;

  f32x4 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool active = false;
  int a0 = 0, a1 = 0, bi_ = 0, kh_ = 0, cw_ = 0;

  if (wave >= kNComp) {
    // ================================================================ loaders
    const int lw = wave - kNComp;
    __builtin_amdgcn_s_setprio(3);   // a handful of SALU + VMEM instructions per chunk: must not queue behind the MFMA stream
    unsigned voff[kMaxNI];
    int my_cnt = 0;
    const unsigned hc_magic = 65536u / (unsigned)HC + 1u;   // hp < 768, HC a multiple of 4 <= 48: exact
#pragma unroll
    for (int i = 0; i < kMaxNI; ++i) {
      const int instr = i * kNLoad + lw;            // instruction `instr` fills LDS pixels [16 instr, 16 instr + 16) of a slot
      voff[i] = 0;
      if (i == 0 ? instr >= nb : instr >= n_instr) continue;   // (wave-uniform)
      const int px = lane >> 2;
      int y, x, key;
      if (i == 0) {                                 // instr < 4: frame-t block `instr` (16 pixels, block-major)
        key = px >> 2;
        y = Y0 + 4 * (instr >> tw_shift) + key;
        x = X0 + 4 * (instr & (tw - 1)) + (px & 3);
      } else {
        const int hp = (instr - 4) * 16 + px, hr = mdiv(hp, hc_magic);
        key = hr & 3;
        y = Y0 - g.R + 4 * r0 + hr;
        x = X0 - g.R + (hp - hr * HC);
      }
      y = min(max(y + g.origin, 0), g.H - 1);       // out-of-image pixels: any in-bounds address (their products are discarded)
      x = min(max(x + g.origin, 0), g.W - 1);
      voff[i] = (unsigned)y * g.sy4 + (unsigned)x * g.sx4 + (unsigned)(((lane & 3) ^ swz(key)) << 4);
      ++my_cnt;
    }
    const char* b1 = reinterpret_cast<const char*>(g.f1 + (long)n * g.sb);
    const char* b2 = reinterpret_cast<const char*>(g.f2 + (long)n * g.sb);
    int slot = 0;
    auto issue = [&](int c) {
      const unsigned dst = lds0 + (unsigned)(slot * slot_bytes + lw * 1024);
      const char* s1 = uptrw(b1 + (long)c * kKC * 4);
      const char* s2 = uptrw(b2 + (long)c * kKC * 4);
      if (!(g.ablate & 1)) {
        if (lw < nb) dma16w(s1, voff[0], dst);
#pragma unroll
        for (int i = 1; i < kMaxNI; ++i)
          if (i * kNLoad + lw < n_instr) dma16w(s2, voff[i], dst + i * kNLoad * 1024);
      }
      slot = slot + 1 == nslot ? 0 : slot + 1;
    };
    if (g.ablate & 1) my_cnt = 0;
    const int ahead = nslot - 2;          // chunk k is read after b_k and b_{k+1}: b_k frees the slot of chunk k-2
    int issued = 0;
    WS_STAMP(1);
    for (; issued < ahead && issued < nch; ++issued) issue(issued);
    // Steady state: when chunk k must have landed, chunks k+1 .. k+ahead-1 (mine: (ahead-1) * my_cnt instructions) may still
    // fly.  s_waitcnt takes an immediate, so the loop is instantiated for a ladder of counts and the allowance is rounded
    // DOWN to the next rung (a stricter wait, never a looser one); the last ahead-1 chunks simply drain (vmcnt(0)).
    const int allow = min((ahead - 1) * my_cnt, 63);
    const int n_steady = max(nch - (ahead - 1), 0);
    auto steady = [&](auto w_tag) {
      constexpr int WAIT = decltype(w_tag)::value;
      for (int k = 0; k < n_steady; ++k) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");   // chunk k has landed (mine)
        wg_barrier();                                     // b_k: chunk k is in LDS for everybody; the slot of chunk k-2 is free
        if (issued < nch) { issue(issued); ++issued; }
      }
    };
#define DTT_RUNG(v) if (allow >= v) steady(std::integral_constant<int, v>{}); else
    DTT_RUNG(48) DTT_RUNG(40) DTT_RUNG(32) DTT_RUNG(28) DTT_RUNG(24) DTT_RUNG(20) DTT_RUNG(16) DTT_RUNG(14) DTT_RUNG(12) DTT_RUNG(10)
    DTT_RUNG(8) DTT_RUNG(7) DTT_RUNG(6) DTT_RUNG(5) DTT_RUNG(4) DTT_RUNG(3) DTT_RUNG(2) DTT_RUNG(1) steady(std::integral_constant<int, 0>{});
#undef DTT_RUNG
    for (int k = n_steady; k < nch; ++k) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wg_barrier();
    }
  } else {
    // ================================================================ compute
    // Two waves per SIMD share a pixel block's window run: wave (cw, kh) takes the chunks k = kh (mod 2).  Every wave passes
    // every barrier; between two barriers one of the pair issues its operand reads (one burst) + half a chunk of MFMAs
    // and the other half a chunk of MFMAs, so a read burst -- which costs a lone MFMA stream ~36 cycles for the first read
    // and ~15 for each further one (tools/probes/mfma_lds.hip) -- always runs underneath the partner's MFMAs.
    const int cw = wave & 3, kh = wave >> 2;
    const int li = lane & 15, lg = lane >> 4, iy = li >> 2, ix = li & 3;
    const int t = 4 * jw + cw, bi = t & (nb - 1), part = t >> nb_shift;
    active = part < g.parts;
    a0 = q0(min(part, g.parts - 1)); a1 = q0(min(part, g.parts - 1) + 1);
    bi_ = bi; kh_ = kh; cw_ = cw;
    const int wy = bi >> tw_shift, wx = bi & (tw - 1);
    const unsigned sw = (unsigned)((lg ^ swz(iy)) << 4);
    const unsigned b_off = (unsigned)((bi * 16 + li) * 64) + sw;
    unsigned a_off[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      const int qb = min(a0 + j, a1 - 1), qi = mdiv(qb, g.nbr_magic), qj = qb - qi * g.nbr;
      a_off[j] = (unsigned)((kPPX + (4 * (wy + qi - r0) + iy) * HC + 4 * (wx + qj) + ix) * 64) + sw;
    }

    if (!active || (g.ablate & 2)) {
      for (int k = 0; k < nch; ++k) wg_barrier();
    } else {
      const char* lb = reinterpret_cast<const char*>(lds);
      // Interval shape (tools/probes/barrier_cost3.hip, cycles per 36-MFMA interval, floor 1152): one wave of the pair
      // issuing all ten reads of its chunk and the other none: 1482; BOTH waves issuing five reads right after the barrier:
      // 1242.  So a wave reads the operands of its chunk in two halves, in two consecutive intervals -- H1 = the frame-t block
      // + the first NLO window blocks after its own barrier, H2 = the other window blocks after the partner's -- and runs the
      // MFMAs of a half one interval after its reads: G1 (NLO blocks x 4 k-steps) then G2.
      constexpr int NLO = NACC / 2, NHI = NACC - NLO;
      f32x4 ALO[NLO > 0 ? NLO : 1], AHI[NHI], BX, BY;
      int slot = kh % nslot;              // slot of my current chunk; chunk k sits in slot k % nslot
      // LDS addresses of my current chunk's operands, made one stage ahead (advance() runs underneath the MFMAs of g1):
      // nothing but the reads themselves stands between a barrier and the MFMAs behind it
      unsigned ab, aa[NACC];
      auto set_addr = [&]() {
        const unsigned cur = (unsigned)(slot * slot_bytes);
        ab = cur + b_off;
#pragma unroll
        for (int j = 0; j < NACC; ++j) aa[j] = cur + a_off[j];
      };
      set_addr();
      auto advance = [&]() {
        slot += 2;
        if (slot >= nslot) slot -= nslot;
        set_addr();
      };
      // (sched_barrier: hipcc otherwise sinks a read burst below the MFMAs of its stage, i.e. right in front of the
      // lgkmcnt(0) of the next barrier, and hoists MFMAs that consume it above that barrier)
      auto rd_h1 = [&](f32x4& B) {
        __builtin_amdgcn_sched_barrier(0);
        B = *reinterpret_cast<const f32x4*>(lb + ab);
#pragma unroll
        for (int j = 0; j < NLO; ++j) ALO[j] = *reinterpret_cast<const f32x4*>(lb + aa[j]);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto rd_h2 = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NHI; ++j) AHI[j] = *reinterpret_cast<const f32x4*>(lb + aa[NLO + j]);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto g1 = [&](const f32x4& B) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NLO; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ALO[j][s], B[s], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto g2 = [&](const f32x4& B) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NHI; ++j) acc[NLO + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(AHI[j][s], B[s], acc[NLO + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto bar = [&]() {
        // my reads of the previous interval are done: b_k frees the slot of chunk k-2.  (The builtin, not inline asm: the
        // compiler's own waitcnt pass then knows that nothing is outstanding and puts no counted s_waitcnt between the
        // MFMAs that consume those registers -- stray issue slots inside an MFMA stream cost tens of cycles each.)
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        wg_barrier();
      };
      WS_STAMP(1);
      for (int k = 0; k < kh && k < nch; ++k) bar();      // the partner's first chunk
      // own chunks kh, kh + 2, ...; own barriers alternate with the partner's; two own chunks per loop iteration (BX / BY)
      const int n_own = nch > kh ? (nch - kh + 1) / 2 : 0;
      const bool extra = n_own > 0 && kh + 2 * (n_own - 1) < nch - 1;   // one more (partner's) barrier after my last chunk
      auto tail = [&](const f32x4& B) {
        if (extra) bar();
        rd_h2(); g1(B); g2(B);
      };
      auto g1a = [&](const f32x4& B) { g1(B); advance(); __builtin_amdgcn_sched_barrier(0); };
      if (n_own > 0) {
        bar(); rd_h1(BX);
        for (int i = 0;;) {
          if (i + 1 >= n_own) { tail(BX); break; }
          bar(); rd_h2(); g1a(BX);
          bar(); rd_h1(BY); g2(BX);
          ++i;
          if (i + 1 >= n_own) { tail(BY); break; }
          bar(); rd_h2(); g1a(BY);
          bar(); rd_h1(BX); g2(BY);
          ++i;
        }
      }
    }
  }

  // ================================================================ epilogue (all waves pass the same barriers)
  WS_STAMP(2);
  __syncthreads();   // E1: every wave is done with the ring
  WS_STAMP(3);
  if (g.ablate & 4) return;
  const bool is_comp = wave < kNComp;
  const int li = lane & 15, lg = lane >> 4;
  // the two chunk phases of a pixel block meet in LDS and share the rest of the work: phase 0 finishes window blocks
  // [0, NLO), phase 1 blocks [NLO, NACC); each hands the other's blocks over (sum = even chunks + odd chunks, fixed order)
  constexpr int NLO_E = NACC / 2;
  float* xbuf = lds + ((long)cw_ * NACC * 64 + lane) * 4;
  if (is_comp && active) {
#pragma unroll
    for (int j = 0; j < NACC; ++j)
      if ((j < NLO_E) != (kh_ == 0)) *reinterpret_cast<f32x4*>(xbuf + j * 256) = acc[j];
  }
  __syncthreads();   // E2
  if (is_comp && active) {
#pragma unroll
    for (int j = 0; j < NACC; ++j)
      if ((j < NLO_E) == (kh_ == 0)) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(xbuf + j * 256);
        acc[j] = kh_ == 0 ? acc[j] + o : o + acc[j];
      }
  }
  __syncthreads();   // E3: the exchange area becomes the output tile
  WS_STAMP(4);
  const int dy_lo = max(-g.R, 4 * r0 - 3 - g.R), dy_hi = min(g.R, 4 * r1 + 3 - g.R), ndy = dy_hi - dy_lo + 1;
  if (is_comp && active) {
    // ---- my window entries into the workgroup's output tile in LDS
    const float inv = 1.f / (float)g.C;
    const int jy = li >> 2, jx = li & 3;                     // D[i][j]: j = lane % 16 the frame-t pixel, i = 4 * (lane / 16) + reg
    const int wy = bi_ >> tw_shift, wx = bi_ & (tw - 1);
    const int py = g.origin + Y0 + 4 * wy + jy, pxx = g.origin + X0 + 4 * wx + jx;
    const bool p_img = py >= 0 && py < g.H && pxx >= 0 && pxx < g.W;
    float* tilebuf = lds + (long)(bi_ * 16 + li) * ndy * g.D;
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if ((j < NLO_E) != (kh_ == 0) || a0 + j >= a1) continue;
      const int qb = a0 + j, qi = mdiv(qb, g.nbr_magic), qj = qb - qi * g.nbr;
      const int dy = 4 * qi + lg - jy - g.R;
      if (dy < -g.R || dy > g.R) continue;
      const int qy = py + dy;
      const bool row_ok = p_img && qy >= 0 && qy < g.H;
      float* trow = tilebuf + (dy - dy_lo) * g.D + g.R;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dx = 4 * qj + r - jx - g.R;
        if (dx < -g.R || dx > g.R) continue;
        const int qx = pxx + dx;
        trow[dx] = (row_ok && qx >= 0 && qx < g.W) ? acc[j][r] * inv : 0.f;
      }
    }
  }
  WS_STAMP(5);
  __syncthreads();     // E4: the tile is assembled
  WS_STAMP(6);

  // ---- stream the tile out in the caller's layout: all 12 waves
  {
    const int per_px = ndy * g.D, npx = 16 * nb;
    float* ob = g.out + (long)n * g.out_sb;
    if (g.out_sc == 1) {
      // position-major rows: the entries of a pixel are one contiguous run of its row, (dy + R) * D + dx + R = e + (dy_lo + R) * D.
      // Which entries of a run belong to THIS workgroup's window parts depends only on the pixel's position inside its 4 x 4
      // block, so a wave takes one in-block position (jy, jx), works the ownership test out once per lane (four entries
      // e = lane + 64 u per round) and then streams that position's pixel of every block of the tile: consecutive lanes
      // write consecutive floats.
      for (int pos = wave; pos < 16; pos += kThreads / 64) {
        const int jy = pos >> 2, jx = pos & 3;
        for (int e0 = lane; e0 < per_px; e0 += 256) {
          bool mine[4];
          int eo[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = e0 + 64 * u;
            const int dyi = mdiv(e, g.d_magic), dxi = e - dyi * g.D;
            const int qb = ((jy + dyi + dy_lo + g.R) >> 2) * g.nbr + ((jx + dxi) >> 2);   // the window block this entry comes from
            mine[u] = e < per_px && qb >= q_lo && qb < q_hi;                               // (else: another workgroup's part)
            eo[u] = min(e, per_px - 1);
          }
          for (int bi = 0; bi < nb; ++bi) {
            const int y = Y0 + 4 * (bi >> tw_shift) + jy, x = X0 + 4 * (bi & (tw - 1)) + jx;
            if (y >= g.oh || x >= g.ow) continue;
            float* orow = ob + ((long)y * g.ow + x) * g.out_sp + (dy_lo + g.R) * g.D;
            const float* trow = lds + (bi * 16 + pos) * per_px;
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = trow[eo[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (mine[u]) orow[eo[u]] = v[u];
          }
        }
      }
    } else {
      // NCHW displacement planes (reference signature, training graph): pixels of a plane side by side
      const int total = npx * per_px;
      const int npx_shift = nb_shift + 4;
      for (int idx = tid; idx < total; idx += kThreads) {
        const int e = idx >> npx_shift, px = idx - (e << npx_shift);
        const int dyi = mdiv(e, g.d_magic), dxi = e - dyi * g.D;
        const int bi = px >> 4, jy = (px >> 2) & 3, jx = px & 3;
        const int y = Y0 + 4 * (bi >> tw_shift) + jy, x = X0 + 4 * (bi & (tw - 1)) + jx;
        if (y >= g.oh || x >= g.ow) continue;
        const int dy = dyi + dy_lo, dx = dxi - g.R;
        const int qb = ((jy + dy + g.R) >> 2) * g.nbr + ((jx + dx + g.R) >> 2);
        if (qb < q_lo || qb >= q_hi) continue;
        ob[(long)((dy + g.R) * g.D + dx + g.R) * g.out_sc + ((long)y * g.ow + x) * g.out_sp] = lds[px * per_px + e];
      }
    }
  }
  WS_STAMP(7);
}

// ------------------------------------------------------------------------------------------------ host side: the plan
constexpr int kNaccSet[] = {3, 5, 7, 9};   // accumulators per wave (13 spills at the 168 registers of 3 waves per SIMD)

struct WPlan {
  int parts, nacc, nbr, nseg, items, nslot, slot_bytes;
  size_t lds_bytes;
  WSeg seg[kMaxSeg];
};

int q0h(int nblk, int parts, int p) { return (nblk * p) / parts; }

// rows of window blocks touched by workgroup j of a tile with nb pixel blocks -> (r0, r1)
void wg_rows(int nblk, int nbr, int parts, int nb, int j, int* r0, int* r1) {
  const int p_lo = (4 * j) / nb, p_hi = std::min(parts - 1, (4 * j + 3) / nb);
  *r0 = q0h(nblk, parts, p_lo) / nbr;
  *r1 = (q0h(nblk, parts, p_hi + 1) - 1) / nbr;
}

// Tiles: 2 x 2 pixel blocks; an odd last block column / row is cut into 4 x 1 / 1 x 4 tiles plus a remainder, so that every
// workgroup but a few has four waves of work (at 10 x 17 blocks: 40 + 2 full tiles and one 2 x 1 per image).
int make_segments(int GH, int GW, WSeg* seg) {
  int ns = 0;
  auto add = [&](int by0, int bx0, int nty, int ntx, int th, int tw) {
    if (nty > 0 && ntx > 0) { seg[ns] = WSeg{0, 0, by0, bx0, nty, ntx, th, tw, 0, 0u, 0u, 0u}; ++ns; }
  };
  const int eh = GH / 2 * 2, ew = GW / 2 * 2;
  add(0, 0, GH / 2, GW / 2, 2, 2);
  if (GW & 1) {
    add(0, ew, eh / 4, 1, 4, 1);
    if (eh % 4) add(eh / 4 * 4, ew, 1, 1, 2, 1);
  }
  if (GH & 1) {
    add(eh, 0, 1, ew / 4, 1, 4);
    if (ew % 4) add(eh, ew / 4 * 4, 1, 1, 1, 2);
  }
  if ((GH & 1) && (GW & 1)) add(eh, ew, 1, 1, 1, 1);
  return ns;
}

int plan_wsplit(int batch, int oh, int ow, int R, int D, int max_wgs, WPlan* best) {
  if (R < 1 || R > 16 || batch < 1) return 0;
  const int nbr = 1 + (R + 1) / 2, nblk = nbr * nbr;
  const int GH = (oh + 3) / 4, GW = (ow + 3) / 4;
  WSeg base[kMaxSeg];
  const int nseg = make_segments(GH, GW, base);
  if (nseg == 0) return 0;
  double best_cost = 1e30;
  bool found = false;
  for (int parts = 1; parts <= nblk; ++parts) {
    const int need = (nblk + parts - 1) / parts;
    int nacc = 0;
    for (int v : kNaccSet) if (v >= need) { nacc = v; break; }
    if (!nacc) continue;
    WPlan p;
    p.parts = parts; p.nacc = nacc; p.nbr = nbr; p.nseg = nseg;
    long items = 0;
    int max_hpx = 0;
    size_t max_tile = 0;
    bool ok = true;
    for (int s = 0; s < nseg && ok; ++s) {
      WSeg sg = base[s];
      const int nb = sg.th * sg.tw;
      sg.wpt = (nb * parts + 3) / 4;
      sg.wpt_magic = 0xffffffffu / (unsigned)sg.wpt + 1u;
      sg.tiles_magic = 0xffffffffu / (unsigned)(sg.nty * sg.ntx) + 1u;
      sg.nty_magic = 0xffffffffu / (unsigned)sg.nty + 1u;
      sg.item0 = (int)items;
      sg.nitems = batch * sg.nty * sg.ntx * sg.wpt;
      items += sg.nitems;
      for (int j = 0; j < sg.wpt; ++j) {
        int r0, r1;
        wg_rows(nblk, nbr, parts, nb, j, &r0, &r1);
        const int hpx = 4 * (r1 - r0 + sg.th) * 4 * (nbr - 1 + sg.tw);
        if ((kPPX + hpx) / 16 > kMaxNI * kNLoad) ok = false;
        max_hpx = std::max(max_hpx, hpx);
        const int ndy = std::min(R, 4 * r1 + 3 - R) - std::max(-R, 4 * r0 - 3 - R) + 1;
        max_tile = std::max(max_tile, (size_t)16 * nb * ndy * D * sizeof(float));
      }
      p.seg[s] = sg;
    }
    if (!ok || items > (1 << 24)) continue;   // (also keeps item * divisor inside 32 bits for the magic divisions)
    p.items = (int)items;
    p.slot_bytes = (kPPX + max_hpx) * kKC * 4;
    p.nslot = std::min(8, kLdsMax / p.slot_bytes);
    if (p.nslot < 3 || max_tile > (size_t)kLdsMax) continue;   // (3 slots: no run-ahead beyond the chunk being awaited -- legal, slow)
    p.lds_bytes = kLdsMax;   // one workgroup per CU (12 waves at 3 per SIMD): the ring takes the whole LDS; the output tile
                             // (max_tile) and the phase exchange (4 * nacc KB) reuse it after the loop
    // time ~ rounds x the slower of a chunk's MFMAs (4 k-steps x nacc x 32 cycles on each SIMD) and its LDS-DMA (pixels x 64 B
    // at kDmaBytesPerClk per CU): finer parts move more halo rows per window block
    const long rounds = (items + max_wgs - 1) / max_wgs;
    const double mfma_clk = 128.0 * nacc, dma_clk = (double)(kPPX + max_hpx) * kKC * 4 / kDmaBytesPerClk;
    const double cost = (double)rounds * std::max(mfma_clk, dma_clk) + 0.5 * parts;
    if (cost < best_cost) { best_cost = cost; *best = p; found = true; }
  }
  return found ? 1 : 0;
}

template <int NACC>
int launch_ws(const WGeom& g, int items, size_t lds_bytes, hipStream_t stream) {
  static DttDeviceOnce once;
  bool& done = once.here();
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_wsplit_kernel<NACC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
    DTT_REQUIRE(e == hipSuccess, "correlation (window-split): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    done = true;
  }
  hipLaunchKernelGGL((corr_wsplit_kernel<NACC>), dim3(items), dim3(kThreads), lds_bytes, stream, g);
  DTT_CHECK_LAUNCH("corr_wsplit_kernel");
  return 1;
}

// the plan's part of the kernel arguments (g.R / g.D set by the caller)
void plan_into_geom(const WPlan& p, WGeom* g) {
  g->nbr = p.nbr; g->nblk = p.nbr * p.nbr; g->parts = p.parts;
  g->nbr_magic = 65536u / (unsigned)p.nbr + 1u;     // x < 81
  g->d_magic = 65536u / (unsigned)g->D + 1u;        // e < 19 * 33
  g->parts_magic = 0xffffffffu / (unsigned)p.parts + 1u;
  g->nseg = p.nseg;
  for (int i = 0; i < kMaxSeg; ++i) g->seg[i] = i < p.nseg ? p.seg[i] : WSeg{0x7fffffff, 0, 0, 0, 1, 1, 1, 1, 1, 0u, 0u, 0u};
  g->ring_bytes = (int)p.lds_bytes;
}

}  // namespace

// input1 / input2: (ob, ih, iw, ic) channels-last.  Output addressing as dtt_correlation_forward_strided.  Supports
// kernel_size 1, stride1 == stride2 = s, displacement and padding multiples of s, ic % 16 == 0, window radius
// max_displacement / s in 1 .. 16; everything else: transpose and call dtt_correlation_forward_strided.
// max_workgroups: 0 = plan one round over every CU; n = plan for n CUs (other kernels run beside this one).
extern "C" int dtt_correlation_forward_nhwc(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                            long out_ch_stride, long out_px_stride, const float* input1, int ic, int ih,
                                            int iw, const float* input2, int pad_size, int kernel_size, int max_displacement,
                                            int stride1, int stride2, int max_workgroups, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(output && input1 && input2, "correlation (channels-last): null pointer");
  int eoc, eoh, eow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &eoc, &eoh, &eow))
    return 0;
  DTT_REQUIRE(ob > 0 && oc == eoc && oh == eoh && ow == eow, "correlation (channels-last): output is (%d,%d,%d,%d), expected (B,%d,%d,%d)",
              ob, oc, oh, ow, eoc, eoh, eow);
  DTT_REQUIRE((((size_t)input1 | (size_t)input2) & 15) == 0, "correlation (channels-last): inputs must be 16-byte aligned");
  DTT_REQUIRE(kernel_size == 1 && stride1 == stride2 && stride1 >= 1, "correlation (channels-last): kernel_size 1 and stride1 == stride2 only");
  const int s = stride1, R = max_displacement / s;
  DTT_REQUIRE(max_displacement % s == 0 && (max_displacement - pad_size) % s == 0,
              "correlation (channels-last): displacement and padding must be multiples of the stride");
  DTT_REQUIRE(ic % kKC == 0, "correlation (channels-last): channels (%d) must be a multiple of %d", ic, kKC);
  DTT_REQUIRE(R >= 1 && R <= 16, "correlation (channels-last): window radius %d not supported (1 .. 16)", R);
  WGeom g;
  g.f1 = input1; g.f2 = input2;
  g.C = ic;
  g.H = (ih + s - 1) / s; g.W = (iw + s - 1) / s;            // lattice pixels 0, s, 2s, ...
  g.sx = (long)s * ic; g.sy = (long)s * iw * ic; g.sb = (long)ih * iw * ic;
  DTT_REQUIRE((long)ih * iw * ic * 4 < 0xffffffffl, "correlation (channels-last): one image exceeds the 32-bit DMA offset range");
  g.sx4 = (unsigned)(g.sx * 4); g.sy4 = (unsigned)(g.sy * 4);
  g.oh = oh; g.ow = ow; g.origin = (max_displacement - pad_size) / s;
  g.R = R; g.D = 2 * R + 1;
  g.out = output; g.out_sb = out_batch_stride; g.out_sc = out_ch_stride; g.out_sp = out_px_stride;
  static const int ablate = getenv("DTT_CORR_WS_ABLATE") ? atoi(getenv("DTT_CORR_WS_ABLATE")) : 0;
  g.ablate = ablate;
  const int ncu = dtt_device_cus();
  const int budget = max_workgroups > 0 ? std::min(max_workgroups, ncu) : ncu;
  WPlan p;
  DTT_REQUIRE(plan_wsplit(ob, oh, ow, R, g.D, budget, &p), "correlation (window-split): no plan for %d x %d outputs, radius %d", oh, ow, R);
  plan_into_geom(p, &g);
  dtt_prof_begin("corr_fwd_op", stream);
  dtt_prof_begin("corr_nhwc", stream);
  int ok = 0;
  switch (p.nacc) {
    case 3: ok = launch_ws<3>(g, p.items, p.lds_bytes, stream); break;
    case 5: ok = launch_ws<5>(g, p.items, p.lds_bytes, stream); break;
    case 7: ok = launch_ws<7>(g, p.items, p.lds_bytes, stream); break;
    default: ok = launch_ws<9>(g, p.items, p.lds_bytes, stream); break;
  }
  dtt_prof_end("corr_nhwc", stream);
  dtt_prof_end("corr_fwd_op", stream);
  return ok;
}

#ifdef DTT_WS_TRACE
extern "C" int dtt_ws_trace_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_ws_trace), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

// Test hook (pure host code): replays EVERY work item of the plan the launcher would use through the kernel's own item decode
// (ws_decode: the same magic-number divisions) and counts, per (image, 4 x 4 pixel block, window block), how many waves own it.
// Returns 1 iff every pair inside the output is owned exactly once, every workgroup's halo fits its DMA budget and at least three
// ring slots, and no wave is given more window blocks than the kernel instantiation holds.
extern "C" int dtt_correlation_nhwc_plan_check(int batch, int oh, int ow, int window_radius, int max_workgroups) {
  WPlan p;
  const int ncu = dtt_device_cus();
  const int budget = max_workgroups > 0 ? std::min(max_workgroups, ncu) : ncu;
  const int R = window_radius;
  if (!plan_wsplit(batch, oh, ow, R, 2 * R + 1, budget, &p)) return 0;
  WGeom g = {};
  g.R = R; g.D = 2 * R + 1;
  plan_into_geom(p, &g);
  const int GH = (oh + 3) / 4, GW = (ow + 3) / 4;
  std::vector<unsigned char> owned((size_t)batch * GH * GW * g.nblk, 0);
  for (int item = 0; item < p.items; ++item) {
    const WItem w = ws_decode(g, item);
    if (w.n < 0 || w.n >= batch || w.q_lo >= w.q_hi || w.r0 > w.r1) return 0;
    const int slot_bytes = (kPPX + w.hrows * w.HC) * kKC * 4;
    if ((kPPX + w.hrows * w.HC) / 16 > kMaxNI * kNLoad || g.ring_bytes / slot_bytes < 3) return 0;
    for (int wave = 0; wave < 4; ++wave) {
      const int t = 4 * w.jw + wave, bi = t & (w.nb - 1), part = t >> w.nb_shift;
      if (part >= g.parts) continue;
      const int a0 = ws_q0(g, part), a1 = ws_q0(g, part + 1);
      if (a1 - a0 > p.nacc || a0 < w.q_lo || a1 > w.q_hi) return 0;
      const int by = w.Y0 / 4 + (bi >> w.tw_shift), bx = w.X0 / 4 + (bi & (w.tw - 1));
      if (by >= GH || bx >= GW) return 0;
      for (int qb = a0; qb < a1; ++qb) {
        if (mdiv(qb, g.nbr_magic) != qb / g.nbr) return 0;
        unsigned char& c = owned[(((size_t)w.n * GH + by) * GW + bx) * g.nblk + qb];
        if (c) return 0;
        c = 1;
      }
    }
  }
  for (unsigned char c : owned)
    if (!c) return 0;
  return 1;
}

// developer / test hook: the plan the launcher would use (parts, accumulators per wave, workgroups, ring slots)
extern "C" int dtt_correlation_nhwc_plan(int batch, int oh, int ow, int window_radius, int max_workgroups, int* parts,
                                         int* accumulators, int* workgroups, int* ring_slots) {
  WPlan p;
  const int ncu = dtt_device_cus();
  const int budget = max_workgroups > 0 ? std::min(max_workgroups, ncu) : ncu;
  if (!plan_wsplit(batch, oh, ow, window_radius, 2 * window_radius + 1, budget, &p)) return 0;
  if (parts) *parts = p.parts;
  if (accumulators) *accumulators = p.nacc;
  if (workgroups) *workgroups = p.items;
  if (ring_slots) *ring_slots = p.nslot;
  return 1;
}

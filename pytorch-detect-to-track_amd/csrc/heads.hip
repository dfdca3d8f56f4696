// R-FCN 1x1 heads on the matrix cores + the position-sensitive pooling that consumes their output (gfx950).
//
// The reference computes the score maps with cuDNN 1x1 convolutions (faster_rcnn/rfcn.py:49-53, 133-140,
// resnet.py:311-312: RFCN_cls_net 512 -> 31*49, RFCN_bbox_net 512 -> 4*49, corr_bbox_net 1051 -> 4*49) into NCHW
// maps whose channel c = (ctop*7 + ph)*7 + pw, and PSROIPoolForward (psroi_pooling_kernel.cu:15-79) then walks one
// channel plane per output bin: neighbouring threads read addresses H*W floats apart.
//
// Here the head is ONE exact-f32 MFMA GEMM over the channels-last trunk output (rows = pixels, K contiguous) that emits
// a POSITION-MAJOR map:  out[pixel][bin*CP + ctop]  with bin = ph*7 + pw and the ctop of one bin contiguous (CP = classes
// padded to a power of two; the weight rows are permuted once on the host).  Pooling then has lanes = classes: every
// element of a bin is ONE aligned 128-byte (31 classes) / 16-byte (4 box deltas) read shared by all the classes of that
// bin, no LDS staging of planes, no bank conflicts, and the 7x7 vote runs in the same workgroup.
//
// head_gemm_kernel  persistent, one workgroup per CU: a strip of TPX*16 pixels x a contiguous range of 16-channel
//                   tiles, processed in `passes` sub-ranges so that the stores of pass p drain underneath the MFMAs of
//                   pass p+1.  Both operands are K-contiguous, so a chunk of 32 k is one 128-byte line per row; rows are
//                   staged global -> LDS by LDS-DMA (global_load_lds_dwordx4) with the 16-byte parts of a row XOR-swizzled
//                   on the SOURCE side (the DMA destination is lane-linear), which makes the operand reads
//                   (one ds_read_b128 = the k-quad of 4 consecutive MFMAs) bank-conflict free.  v_mfma_f32_16x16x4_f32 is
//                   bitwise an fp32 fma chain: no precision is traded.  A = weights (rows -> 4 consecutive channels per
//                   lane in the accumulator), B = pixels, so the epilogue is one 16-byte store per accumulator.
// psroi_pm_kernel   one workgroup per RoI; lane = (bin slot, class).  Bin geometry and summation order are the
//                   reference's, so pooled bins and votes are bit-identical to the oracle's on the same map.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "psroi_bin.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

constexpr int kBK = 32;   // k per stage = one 128-byte line per row

// The wide / narrow (non-RPN) epilogue stores with the non-temporal hint.  Round 5, rocprofv3 --pmc WRITE_SIZE on the class + box launch
// (10184 x 1764 floats = 71.9 MB stored): plain 16-byte stores 92.2 MB (1.28x: the 64-byte half lines of an accumulator tile are written
// back by the L2 more than once), non-temporal 72.2 MB (1.005x), the same duration (the launch is MFMA-bound).  A transposed product
// (A = pixels: 64 contiguous bytes per quarter-wave, four 4-byte stores per accumulator) left the counter at 93.7 MB -- the lane pattern
// was not the cause (profiles/r05_head_store_ab.txt).  -DDTT_HEAD_PLAIN_STORES restores the plain stores for an A/B.

constexpr int kMaxPasses = 8;

#ifdef DTT_HEAD_STAMP   // developer timeline: shader-clock stamps of workgroup 0 (tools/head_timeline.py)
__device__ unsigned long long dtt_head_stamps[8 * 512];
#define HEAD_STAMP(w, idx) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (idx) < 512) dtt_head_stamps[(w) * 512 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define HEAD_STAMP(w, idx) do {} while (0)
#endif

struct HeadGeom {
  const float* x; long ldx;   // (M, K) pixel rows (channels-last), ldx floats between rows
  const float* w;             // (nt_total*16, K): output channels in emitted order, zero rows as padding
  const float* bias;          // nt_total*16
  float* out; long ldc;       // (M, ldc); columns >= n_store are never written
  int M, K, n_store;
  int nt_total, n_groups, strips, passes;
  int pass_len[kMaxPasses];   // channel tiles per pass (the last pass takes what is left of the group)
  int ablate;                 // developer timing experiments (DTT_HEAD_ABLATE): 1 no DMA, 2 no MFMA, 4 no stores, 8 no DMA wait, 16 no X DMA, 32 no W DMA
  // epilogue 1 (the RPN's heads, rpn.py:63-71): emitted channels [0, 2A) are (background, foreground) score PAIRS of anchor
  // a = channel / 2 -> softmax over the pair -> planes a and A + a of `out` = rpn_cls_prob (B, 2A, H, W); emitted channels
  // [2A, 6A) -> planes of `out2` = rpn_bbox_pred (B, 4A, H, W).  Rows are pixels, image-major: row = b * hw + p.
  int epilogue, A, hw;
  unsigned hw_magic;          // (2^32 - 1) / hw + 1
  float* out2;
};

// One LDS-DMA instruction in the scalar-base form: 16 bytes per lane from sbase + voff (per-lane byte offset) to LDS byte
// address lds_addr + lane * 16.  Written as asm so that the loader's instruction stream is SALU + VMEM only: a VALU
// address add would queue behind the MFMAs of the compute wave that shares the SIMD (see head_gemm_kernel).
__device__ __forceinline__ void dma16(const char* sbase, unsigned voff, unsigned lds_addr) {
  // m0 (the DMA's LDS base) is put back inside the statement: it is a reserved register the compiler neither allocates nor saves
  // around inline asm -- a statement that merely listed it as clobbered would rely on hipcc never keeping a value of its own there
  unsigned keep_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep_m0)
               : "s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  return (const char*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

__device__ __forceinline__ int part_begin(int n, int k, int i) { return (int)((unsigned)(n * i) / (unsigned)k); }

// first tile (relative to the group) of pass p, p in [0, passes]
__device__ __forceinline__ int pass_begin(const HeadGeom& g, int ntg, int p) {
  if (p >= g.passes) return ntg;
  int b = 0;
  for (int q = 0; q < p; ++q) b += g.pass_len[q];
  return min(b, ntg);
}

// Compute side of one pass (= one sub-range of the workgroup's channel tiles, all K) with a compile-time tile count
// per wave: the accumulators sit in fixed registers for the whole K loop.  The compute waves never issue a load from
// global memory, so nothing ever makes them wait for their own stores: the tile of pass p drains underneath pass p+1.
//
// A chunk (32 k) is 2 * NG segments = (16-k half) x (group of TPG pixel tiles); a segment is CNT blocks of 4 * TPG MFMAs
// (one weight fragment against the group's pixel fragments).  Operands are requested one block (weights) / one segment
// (pixels) ahead of the MFMAs that consume them, and the barrier that publishes chunk s+1 sits BEFORE the last segment of
// chunk s: the first operands of chunk s+1 are fetched underneath that segment, so the matrix pipe does not drain at
// chunk boundaries.  Three LDS stages make that legal (after barrier s+1 the loaders fill stage s+2 while stages s and
// s+1 are both still being read).
template <int TPX, int NTW, int CNT, bool PIN, int NSTAGE, bool RPN>
__device__ __forceinline__ void head_pass(const HeadGeom& g, const float* lds, int s_begin, int tile0, int t0, int cnt,
                                          int p0, int lane, int wave) {
  constexpr int XROWS = TPX * 16;
  constexpr int STAGE = (XROWS + 4 * NTW * 16) * kBK;
  constexpr int NG = (TPX % 2 == 0 && TPX > 5) ? 2 : 1, TPG = TPX / NG, NSEG = 2 * NG;
  const int KC = g.K / kBK;
  const int l15 = lane & 15, lg = lane >> 4;
  const int sw[2] = {(lg ^ (lane & 7)) * 4, ((4 + lg) ^ (lane & 7)) * 4};   // swizzled k-quad offsets, two 16-k halves
  // the accumulators start from the bias (read while the first chunk is still landing): nothing but stores at the end
  f32x4 acc[CNT][TPX];
#pragma unroll
  for (int t = 0; t < CNT; ++t) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + (long)min(tile0 + t, g.nt_total - 1) * 16 + 4 * lg);
#pragma unroll
    for (int pt = 0; pt < TPX; ++pt) acc[t][pt] = b4;
  }
  // every wave runs CNT tiles (a wave that owns fewer recomputes a neighbour's: the barrier makes the step as long as
  // its slowest wave anyway, and the extra tile is never stored)
  const int xrow = l15 * kBK;
  const int wrow = (XROWS + min(t0, 4 * NTW - CNT) * 16 + l15) * kBK;
  auto stage_of = [&](int s) { return lds + (s % NSTAGE) * STAGE; };
  auto read_bx = [&](f32x4 (&dst)[TPG], const float* st, int seg) {
    const int hh = seg / NG, pxg = seg % NG;
#pragma unroll
    for (int q = 0; q < TPG; ++q) dst[q] = *reinterpret_cast<const f32x4*>(st + xrow + (pxg * TPG + q) * 16 * kBK + sw[hh]);
  };
  auto read_a = [&](const float* st, int hh, int t) { return *reinterpret_cast<const f32x4*>(st + wrow + t * 16 * kBK + sw[hh]); };

  // Operand traffic is batched: ONE burst of LDS reads per segment fetches everything the NEXT segment needs (its pixel
  // fragments, and the CNT weight fragments when the 16-k half changes).  A lone ds_read between two MFMAs costs the
  // single-wave MFMA stream ~36 cycles whatever its width, further reads in the same burst ~15 each
  // (tools/probes/mfma_lds.hip), so fewer, fatter interruptions win.
  f32x4 bx[2][TPG], aw[2][CNT];
  HEAD_STAMP(wave, 2 * s_begin);
  __builtin_amdgcn_s_barrier();   // the first chunk of the pass has landed
  HEAD_STAMP(wave, 2 * s_begin + 1);
  if (!(g.ablate & 2)) {
    read_bx(bx[0], stage_of(s_begin), 0);
#pragma unroll
    for (int t = 0; t < CNT; ++t) aw[0][t] = read_a(stage_of(s_begin), 0, t);
  }
  auto do_step = [&](auto last_tag, int kc) {
    constexpr bool LAST = decltype(last_tag)::value;   // last chunk of the pass: nothing to publish or prefetch
    const float* cur = stage_of(s_begin + kc);
    const float* nxt = stage_of(s_begin + kc + 1);
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
      const int hh = seg / NG, pxg = seg % NG;
      if (seg == NSEG - 1 && !LAST) {
        HEAD_STAMP(wave, 2 * (s_begin + kc + 1));
        __builtin_amdgcn_s_barrier();   // chunk kc+1 has landed; the loaders move on to kc+2
        HEAD_STAMP(wave, 2 * (s_begin + kc + 1) + 1);
      }
      if (g.ablate & 2) continue;
      if (seg < NSEG - 1) {
        read_bx(bx[(seg + 1) & 1], cur, seg + 1);
        if ((seg + 1) / NG != hh) {
#pragma unroll
          for (int t = 0; t < CNT; ++t) aw[1][t] = read_a(cur, 1, t);
        }
      } else if (!LAST) {
        read_bx(bx[0], nxt, 0);
#pragma unroll
        for (int t = 0; t < CNT; ++t) aw[0][t] = read_a(nxt, 0, t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < CNT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int q = 0; q < TPG; ++q)
            acc[t][pxg * TPG + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[hh][t][j], bx[seg & 1][q][j], acc[t][pxg * TPG + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto pin = [&]() {   // keep the accumulators in the AGPR half of the register file: the MFMA then reads A / B from
                       // VGPR banks and C from AGPR banks (VGPR-resident accumulators cost ~5 cycles per MFMA in operand fetch)
#pragma unroll
    for (int t = 0; t < CNT; ++t)
#pragma unroll
      for (int pt = 0; pt < TPX; ++pt) if constexpr (PIN) asm volatile("" : "+a"(acc[t][pt]));
  };
  pin();
  for (int kc = 0; kc + 1 < KC; ++kc) { do_step(std::false_type{}, kc); pin(); }
  do_step(std::true_type{}, KC - 1);
  HEAD_STAMP(wave, 400 + 4 * (s_begin / KC));
  if constexpr (RPN) {
    // RPN heads: pairwise softmax in registers (a lane holds two whole (bg, fg) pairs), NCHW planes out -- the layout
    // proposal_select_sort / proposal_decode read; 16 consecutive pixels per lane group = 64-byte runs
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
      if (t < cnt && !(g.ablate & 4)) {
        const int col = (tile0 + t) * 16 + 4 * lg;
#pragma unroll
        for (int pt = 0; pt < TPX; ++pt) {
          const int row = p0 + pt * 16 + l15;
          if (row >= g.M || col >= g.n_store) continue;
          const int b = (int)__umulhi((unsigned)row, g.hw_magic), p = row - b * g.hw;
          const f32x4 v = acc[t][pt];
          if (col < 2 * g.A) {
            float* o = g.out + ((long)b * 2 * g.A) * g.hw + p;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int a = (col >> 1) + h;
              const float m = fmaxf(v[2 * h], v[2 * h + 1]);
              const float e0 = expf(v[2 * h] - m), e1 = expf(v[2 * h + 1] - m), inv = 1.f / (e0 + e1);
              if (a < g.A) {
                o[(long)a * g.hw] = e0 * inv;
                o[(long)(g.A + a) * g.hw] = e1 * inv;
              }
            }
          } else {
            float* o = g.out2 + ((long)b * 4 * g.A + (col - 2 * g.A)) * g.hw + p;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (col - 2 * g.A + r < 4 * g.A) o[(long)r * g.hw] = v[r];
          }
        }
      }
    }
  } else {
    // (one wave-uniform base + 32-bit per-lane byte offsets, M * ldc * 4 < 2^32 is checked by the host: forty 64-bit row pointers
    //  per lane were what the 256-register wide configuration spilled)
    char* ob = reinterpret_cast<char*>(g.out);
    const unsigned ldc4 = (unsigned)(g.ldc * 4);
    const unsigned row0 = (unsigned)(p0 + l15);
#pragma unroll
    for (int t = 0; t < CNT; ++t) {
      if (t < cnt && !(g.ablate & 4)) {
        const int col = (tile0 + t) * 16 + 4 * lg;
        const unsigned off0 = row0 * ldc4 + (unsigned)col * 4u;
#pragma unroll
        for (int pt = 0; pt < TPX; ++pt) {
          const int row = p0 + pt * 16 + l15;
          if (row < g.M && col < g.n_store) {
#ifdef DTT_HEAD_PLAIN_STORES
            *reinterpret_cast<f32x4*>(ob + (off0 + (unsigned)(pt * 16) * ldc4)) = acc[t][pt];
#else
            __builtin_nontemporal_store(acc[t][pt], reinterpret_cast<f32x4*>(ob + (off0 + (unsigned)(pt * 16) * ldc4)));
#endif
          }
        }
      }
    }
  }
  HEAD_STAMP(wave, 401 + 4 * (s_begin / KC));
}

// 4 compute waves (one per SIMD) + NLOAD loader waves.  Loaders move chunk s+1 global -> LDS by LDS-DMA while the compute
// waves run the MFMAs of chunk s; one barrier per chunk, three LDS stages (see head_pass).
template <int TPX, int NTW, int NLOAD, bool PIN, int NSTAGE, bool RPN>
__global__ __launch_bounds__((4 + NLOAD) * 64) void head_gemm_kernel(HeadGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int XROWS = TPX * 16;
  constexpr int STAGE = (XROWS + 4 * NTW * 16) * kBK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  // ablate & 64: channel groups of a strip on different XCDs (group-major order) instead of side by side on one XCD
  const int grp = (g.ablate & 64) ? item / g.strips : item % g.n_groups;
  const int strip = (g.ablate & 64) ? item - grp * g.strips : item / g.n_groups;
  const int p0 = strip * XROWS;
  const int nt_lo = part_begin(g.nt_total, g.n_groups, grp);
  const int ntg = part_begin(g.nt_total, g.n_groups, grp + 1) - nt_lo;
  const int KC = g.K / kBK;

  if (wave >= 4) {
    // ---- loader: per chunk, the X rows of the strip and the W rows of the pass's tiles; 8 rows (1 KB) per instruction,
    // the 16-byte parts of a row XOR-swizzled on the source side (the LDS image of an instruction is lane-linear).
    // Instruction i of a chunk belongs to loader i % NLOAD.  Per-lane byte offsets are fixed for the whole launch; only
    // the wave-uniform bases move (k0 per chunk, the W row block per pass).
    static_assert((XROWS / 8) % NLOAD == 0, "X instructions split evenly over the loaders");
    constexpr int NX = XROWS / 8 / NLOAD;
    const int lw = wave - 4;
    // The loader shares its SIMD with a compute wave that always has an MFMA ready to issue; at equal priority the older
    // (compute) wave wins every arbitration and the DMA instructions only get out while the compute waves sit at the
    // barrier -- one whole chunk late.  A handful of high-priority instructions per chunk cost the MFMA stream nothing.
    if (!(g.ablate & 128)) __builtin_amdgcn_s_setprio(3);
    const int r8 = lane >> 3;
    const unsigned part = (unsigned)(((lane & 7) ^ r8) << 4);   // bytes
    unsigned xoff[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) xoff[j] = (unsigned)min(p0 + (j * NLOAD + lw) * 8 + r8, g.M - 1) * (unsigned)(g.ldx * 4) + part;
    const unsigned woff = (unsigned)(lw * 8 + r8) * (unsigned)(g.K * 4) + part;
    const unsigned wstep = (unsigned)(8 * NLOAD) * (unsigned)(g.K * 4);
    const char* xb = reinterpret_cast<const char*>(g.x);
    const unsigned lds_base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)lds;
    int s = 0;
    for (int pass = 0; pass < g.passes; ++pass) {
      const int pb = pass_begin(g, ntg, pass), L = pass_begin(g, ntg, pass + 1) - pb;
      const int nw = (2 * L - lw + NLOAD - 1) / NLOAD;    // this loader's W instructions per chunk
      const char* wb = reinterpret_cast<const char*>(g.w + (long)(nt_lo + pb) * 16 * g.K);
      // one straight-line burst of DMA instructions per chunk (the loader shares its SIMD with a compute wave: the
      // fewer instructions it issues, the fewer times the MFMA stream next to it is interrupted)
      auto chunk = [&](auto nw_tag) {
        constexpr int NWI = decltype(nw_tag)::value;
        for (int kc = 0; kc < KC; ++kc, ++s) {
          const unsigned stage = lds_base + (unsigned)((s % NSTAGE) * STAGE * 4 + lw * 1024);
          if (!(g.ablate & 1)) {
            const char* xs = uniform_ptr(xb);
#pragma unroll
            for (int j = 0; j < NX; ++j)
              if (!(g.ablate & 16)) dma16(xs, xoff[j], stage + j * NLOAD * 1024);
            if constexpr (NWI >= 0) {
#pragma unroll
              for (int i = 0; i < NWI; ++i)
                if (!(g.ablate & 32)) dma16(uniform_ptr(wb + (size_t)i * wstep), woff, stage + (XROWS / 8 + i * NLOAD) * 1024);
            } else {   // more instructions per chunk than the unrolled variants cover
              const char* ws = wb;
              unsigned dst = stage + XROWS / 8 * 1024;
              for (int i = 0; i < nw; ++i, ws += wstep, dst += NLOAD * 1024) dma16(uniform_ptr(ws), woff, dst);
            }
          }
          xb += kBK * 4;
          wb += kBK * 4;
          HEAD_STAMP(wave, 3 * s);
          if constexpr (NSTAGE > 3) {
            // a fourth stage lets the loader run one chunk ahead: chunk s stays in flight while chunk s-1 is published (barrier
            // #(s-1) comes one iteration late; the last one after the loop).  Narrow heads have so few MFMAs per chunk that the
            // DMA latency of every chunk was exposed: 16 chunks x ~1.3 us for the RPN's heads.
            if (s > 0) {
              if constexpr (NWI >= 0) { if (!(g.ablate & 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NX + NWI) : "memory"); }
              else { if (!(g.ablate & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
              HEAD_STAMP(wave, 3 * s + 1);
              __builtin_amdgcn_s_barrier();   // chunk s-1 is in LDS; every compute wave has finished chunk s-3
            }
          } else {
            if (!(g.ablate & 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            HEAD_STAMP(wave, 3 * s + 1);
            __builtin_amdgcn_s_barrier();   // chunk s is in LDS; every compute wave has finished chunk s-2 (the stage refilled next)
          }
          HEAD_STAMP(wave, 3 * s + 2);
        }
      };
      switch (nw) {
        case 0: chunk(std::integral_constant<int, 0>{}); break;
        case 1: chunk(std::integral_constant<int, 1>{}); break;
        case 2: chunk(std::integral_constant<int, 2>{}); break;
        case 3: chunk(std::integral_constant<int, 3>{}); break;
        case 4: chunk(std::integral_constant<int, 4>{}); break;
        case 5: chunk(std::integral_constant<int, 5>{}); break;
        case 6: chunk(std::integral_constant<int, 6>{}); break;
        case 7: chunk(std::integral_constant<int, 7>{}); break;
        case 8: chunk(std::integral_constant<int, 8>{}); break;
        default: chunk(std::integral_constant<int, -1>{}); break;
      }
      xb -= (long)KC * kBK * 4;
    }
    if constexpr (NSTAGE > 3) {
      if (s > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // the last chunk
      }
    }
    return;
  }

  HEAD_STAMP(wave, 500);
  int s_begin = 0;
  for (int pass = 0; pass < g.passes; ++pass, s_begin += KC) {
    const int pb = pass_begin(g, ntg, pass), L = pass_begin(g, ntg, pass + 1) - pb;
    const int cmax = (L + 3) >> 2;   // workgroup-uniform tiles per wave in this pass
    const int t0 = part_begin(L, 4, wave), cnt = part_begin(L, 4, wave + 1) - t0;
    const int tile0 = nt_lo + pb + t0;
#define DTT_HEAD_PASS(C) head_pass<TPX, NTW, C, PIN, NSTAGE, RPN>(g, lds, s_begin, tile0, t0, cnt, p0, lane, wave)
    if constexpr (NTW >= 7) { if (cmax == 7) { DTT_HEAD_PASS(7); continue; } }
    if constexpr (NTW >= 6) { if (cmax == 6) { DTT_HEAD_PASS(6); continue; } }
    if constexpr (NTW >= 5) { if (cmax == 5) { DTT_HEAD_PASS(5); continue; } }
    if constexpr (NTW >= 4) { if (cmax == 4) { DTT_HEAD_PASS(4); continue; } }
    if constexpr (NTW >= 3) { if (cmax == 3) { DTT_HEAD_PASS(3); continue; } }
    if constexpr (NTW >= 2) { if (cmax == 2) { DTT_HEAD_PASS(2); continue; } }
    DTT_HEAD_PASS(1);
#undef DTT_HEAD_PASS
  }
}

// ------------------------------------------------------------------------------------------------ pooling
// One workgroup per RoI.  A lane owns VPL = CP / LPB consecutive classes of ONE bin (16-byte loads); the LPB neighbouring
// lanes cover the classes of that bin, so a wave instruction gathers whole 128-byte (31 classes) / 16-byte (4 box deltas)
// runs of 64 / LPB different bins at once and nothing is ever reduced across lanes: every (bin, class) sum is one lane's
// sequential walk over the bin in the reference's (h, w) order, with kFlight positions in flight.  map(b, h, w, bin, c) =
// map[((b*H + h)*W + w) * pixel_stride + bin*CP + c].  Bins land in LDS [bin][CP]; the vote (rfcn.py:62-64: AvgPool2d over
// the P x P bins) is the reference's row-major sum followed by one division; `pooled_out` (optional) receives the bins in
// the reference layout (R, od, P, P).  All RoIs are resident at once (4 waves per RoI at the 31-class shape).
// The bins of one RoI for one head: thread `t` of `nthreads` owns VPL = CP / LPB classes of the bins slot, slot + SLOTS, ...
// (h, w) row-major walk, kFlight positions in flight, adds in order: every (bin, class) sum is one lane's sequential walk in the
// reference's order.  bins_out: LDS [pooled * pooled][CP].
template <int CP, int LPB, int kFlight = 8>
__device__ __forceinline__ void pm_pool_bins(const float* __restrict__ img, long pixel_stride, int height, int width, const float* roi,
                                             float spatial_scale, int pooled, int t, int nthreads, float* __restrict__ bins_out) {
  constexpr int VPL = CP / LPB, NV = VPL / 4;   // classes / 16-byte pieces per lane; kFlight positions of a bin in flight per lane
  const int SLOTS = nthreads / LPB;             // bins in flight
  const int cq = t % LPB, slot = t / LPB;
  const int nbins = pooled * pooled;
  img += cq * VPL;
  for (int bin = slot; bin < nbins; bin += SLOTS) {
    const int ph = bin / pooled, pw = bin - ph * pooled;
    const Bin g = psroi_bin(roi, spatial_scale, ph, pw, pooled, pooled, height, width);
    f32x4 sum[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!g.empty) {
      const int nw = g.wend - g.wstart, area = (g.hend - g.hstart) * nw;
      const float* p = img + (long)bin * CP;
      int h = g.hstart, w = g.wstart;
      for (int i = 0; i < area; i += kFlight) {
        f32x4 v[kFlight][NV];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
          if (i + u < area) {
            const float* src = p + ((long)h * width + w) * pixel_stride;
#pragma unroll
            for (int q = 0; q < NV; ++q) v[u][q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
          }
          if (++w == g.wend) { w = g.wstart; ++h; }
        }
#pragma unroll
        for (int u = 0; u < kFlight; ++u)
          if (i + u < area) {
#pragma unroll
            for (int q = 0; q < NV; ++q) sum[q] += v[u][q];
          }
      }
      const float fa = (float)area;
#pragma unroll
      for (int q = 0; q < NV; ++q) sum[q] = f32x4{sum[q][0] / fa, sum[q][1] / fa, sum[q][2] / fa, sum[q][3] / fa};
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) *reinterpret_cast<f32x4*>(&bins_out[bin * CP + cq * VPL + 4 * q]) = sum[q];
  }
}

// the vote of class `c` (rfcn.py:62-64: AvgPool2d over the P x P bins): the reference's row-major sum, one division
template <int CP, int POOLED>
__device__ __forceinline__ float pm_vote(const float* __restrict__ bins, int c, int nbins) {
  float s = 0.f;
  if (POOLED > 0) {       // all loads first, then the adds in order
    float v[POOLED > 0 ? POOLED * POOLED : 1];
#pragma unroll
    for (int k = 0; k < POOLED * POOLED; ++k) v[k] = bins[k * CP + c];
#pragma unroll
    for (int k = 0; k < POOLED * POOLED; ++k) s += v[k];
  } else {
    for (int k = 0; k < nbins; ++k) s += bins[k * CP + c];
  }
  return s / (float)nbins;
}

template <int CP, int LPB, int NW, int POOLED>
__global__ __launch_bounds__(NW * 64) void psroi_pm_kernel(const float* __restrict__ map, long pixel_stride, int height,
                                                           int width, const float* __restrict__ rois,
                                                           float spatial_scale, int pooled_rt, int output_dim,
                                                           float* __restrict__ vote, float* __restrict__ pooled_out, int batch_size) {
  extern __shared__ __attribute__((aligned(16))) float bins[];   // [pooled*pooled][CP]
  const int pooled = POOLED > 0 ? POOLED : pooled_rt;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  float roi[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) roi[q] = rois[(long)n * 5 + q];
  // a batch index outside the map (a caller's bug: the reference reads out of bounds there) pools image 0 instead of wild memory
  const int b = min(max((int)roi[0], 0), batch_size - 1);
  const int nbins = pooled * pooled;
  pm_pool_bins<CP, LPB>(map + (long)b * height * width * pixel_stride, pixel_stride, height, width, roi, spatial_scale, pooled, tid, NW * 64, bins);
  __syncthreads();
  if (tid < output_dim) vote[(long)n * output_dim + tid] = pm_vote<CP, POOLED>(bins, tid, nbins);   // one thread per class
  if (pooled_out) {
    float* o = pooled_out + (long)n * output_dim * nbins;
    for (int i = tid; i < output_dim * nbins; i += NW * 64) {
      const int ct = i / nbins, k = i - ct * nbins;
      o[i] = bins[k * CP + ct];
    }
  }
}

// Waves per SIMD the detection pooling is compiled for (the second __launch_bounds__ argument) and positions in flight per lane of
// the class part.  A workgroup is 5 waves; at 86 registers 5 waves fit a SIMD, i.e. 4 workgroups a CU: 1024 of the 1200 RoIs of an
// inference step are resident at once.  Round 5 A/B inside bench.py (profiles/r05_psroi_det_ab.txt): 7 waves x 6 in flight (68
// registers, 1280 RoIs resident) 28.1 us, 8 x 5 29.0, 8 x 4 31.9 against 28.9 for 5 x 8 on the same box -- residency is not what bounds
// the launch, the per-lane chain of dependent round trips over a large RoI's bins is (fewer positions in flight = more trips).
#ifndef DTT_PSROI_DET_WAVES
#define DTT_PSROI_DET_WAVES 5
#endif
#ifndef DTT_PSROI_DET_FLIGHT
#define DTT_PSROI_DET_FLIGHT 8
#endif

// Detection pooling of a RoI in ONE launch (rfcn.py:133-140): the class scores (CP = 32 slots per bin, waves 0 - 3) and the box
// deltas (4 per bin, wave 4: lane = bin) of the same position-major map, the same bin edges, the same rows -- and the softmax over
// the classes folded into the epilogue (the 31 votes of a RoI sit in one wave).  Votes are the very sums of psroi_pm_kernel
// (shared code); cls_prob = exp(s - max) / sum exp(s - max), with the sum taken in class order.
template <int POOLED>
__global__ __launch_bounds__(320, DTT_PSROI_DET_WAVES) void psroi_pm_det_kernel(const float* __restrict__ map, long pixel_stride, int loc_offset, int height,
                                                           int width, const float* __restrict__ rois, float spatial_scale, int pooled_rt,
                                                           int n_cls, int n_loc, float* __restrict__ cls_vote, float* __restrict__ cls_prob,
                                                           float* __restrict__ loc_vote, int batch_size) {
  extern __shared__ __attribute__((aligned(16))) float bins[];   // [pooled*pooled][32] class bins, then [pooled*pooled][4] box bins
  const int pooled = POOLED > 0 ? POOLED : pooled_rt;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  float roi[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) roi[q] = rois[(long)n * 5 + q];
  const int b = min(max((int)roi[0], 0), batch_size - 1);
  const int nbins = pooled * pooled;
  const float* img = map + (long)b * height * width * pixel_stride;
  float* lbins = bins + nbins * 32;
  // (class part: DTT_PSROI_DET_FLIGHT positions x two 16-byte pieces in flight per lane -- what fits the register budget above)
  if (tid < 256) pm_pool_bins<32, 4, DTT_PSROI_DET_FLIGHT>(img, pixel_stride, height, width, roi, spatial_scale, pooled, tid, 256, bins);
  else pm_pool_bins<4, 1>(img + loc_offset, pixel_stride, height, width, roi, spatial_scale, pooled, tid - 256, 64, lbins);
  __syncthreads();
  if (tid < 64) {           // wave 0: the class votes and their softmax
    const float s = tid < n_cls ? pm_vote<32, POOLED>(bins, tid, nbins) : -3.0e38f;
    if (tid < n_cls && cls_vote) cls_vote[(long)n * n_cls + tid] = s;
    float m = s;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    const float e = tid < n_cls ? expf(s - m) : 0.f;
    float sum = 0.f;
    for (int c = 0; c < n_cls; ++c) sum += __shfl(e, c, 64);     // (class order: the same sum on every lane)
    if (tid < n_cls) cls_prob[(long)n * n_cls + tid] = e / sum;
  } else if (tid >= 256 && tid - 256 < n_loc) {
    loc_vote[(long)n * n_loc + (tid - 256)] = pm_vote<4, POOLED>(lbins, tid - 256, nbins);
  }
}

// ------------------------------------------------------------------------------------------------ pooling, backward
// Gradient of psroi_pm_kernel's vote with respect to the position-major map (PSROIPoolBackward, psroi_pooling_kernel.cu:109-170,
// composed with the AvgPool2d of rfcn.py:62-64):  d map[b, h, w, bin, c] = sum over the RoIs r of image b whose bin `bin`
// contains (h, w) of  gvote[r, c] / (P * P) / area(r, bin).  The reference scatters with atomicAdd (summation order = whatever
// the hardware does); here the map is stationary: one workgroup per PIXEL collects, in RoI order, the (RoI, bin) pairs that
// cover it (ballot-free ordered compaction through an LDS prefix sum) and one wave per bin subset adds them in that order,
// lanes = classes -- no atomics, no pre-zeroed output (every pixel writes its whole row segment), deterministic.
// psroi_pm_edges_kernel first turns every RoI into its 4 * P bin edges with the forward's arithmetic (psroi_bin.h).
__global__ void psroi_pm_edges_kernel(const float* __restrict__ rois, int num_rois, float spatial_scale, int pooled, int height,
                                      int width, int batch_size, int* __restrict__ edges, unsigned* __restrict__ range,
                                      const float* __restrict__ gvote, int output_dim) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= num_rois) return;
  float roi[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) roi[q] = rois[(long)r * 5 + q];
  // word k of RoI r at edges[k * num_rois + r]: the pixel workgroups read word k of 256 consecutive RoIs at a time (the first
  // version kept a RoI's 4 P + 1 words together: 64 cache lines per wave load in the per-pixel search, 73 M line requests per call)
  int* e = edges + r;
  const long ns = num_rois;
  int b = min(max((int)roi[0], 0), batch_size - 1);           // the forward's clamp (psroi_pm_kernel): both directions agree on the image
  // A RoI whose gradient row is all zeros adds nothing anywhere (background RoIs in the box head: three quarters of the sampled
  // RoIs; the zero-padded ground-truth rows of the tracking RoIs -- (0,0,0,0) boxes whose 49 bins ALL cover pixel (0, 0): 1400
  // hits walked by one workgroup, the whole launch waiting for it): it is given to no image
  bool nonzero = false;
  for (int c = 0; c < output_dim; ++c) nonzero |= gvote[(long)r * output_dim + c] != 0.f;
  if (!nonzero) b = -1;
  e[4 * pooled * ns] = b;
  if (b < 0) return;
  // the RoIs of image b lie in [first, last] (pre-zeroed words: ~first and last + 1 by atomicMax): a pixel's workgroup only walks
  // that run -- callers list their RoIs image by image, so the run is the image's own RoIs
  atomicMax(&range[2 * b], ~(unsigned)r);
  atomicMax(&range[2 * b + 1], (unsigned)r + 1u);
  for (int k = 0; k < pooled; ++k) {
    const Bin bn = psroi_bin(roi, spatial_scale, k, k, pooled, pooled, height, width);   // rows depend on ph only, columns on pw only
    e[k * ns] = bn.hstart; e[(pooled + k) * ns] = bn.hend; e[(2 * pooled + k) * ns] = bn.wstart; e[(3 * pooled + k) * ns] = bn.wend;
  }
}

template <int CP, int POOLED>
__global__ __launch_bounds__(256) void psroi_pm_bwd_kernel(const float* __restrict__ gvote, const int* __restrict__ edges, int num_rois,
                                                           int output_dim, int pooled_rt, int height, int width, long pixel_stride,
                                                           float* __restrict__ gmap, const unsigned* __restrict__ range) {
  const int pooled = POOLED > 0 ? POOLED : pooled_rt;   // (compile-time 7: the edge words of a RoI are requested in one batch)
  constexpr int kMaxHits = 2048;                       // (RoI, bin, weight) triples per chunk of 256 RoIs: at most 4 x 4 bins each in theory,
  __shared__ int hit_r[kMaxHits];                      // 2 x 2 in practice; a chunk that overflows is split (see below)
  __shared__ short hit_bin[kMaxHits];
  __shared__ float hit_w[kMaxHits];
  __shared__ int wave_cnt[4];
  extern __shared__ __attribute__((aligned(16))) float accum[];   // [pooled*pooled][CP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = blockIdx.x;
  const int hw = height * width;
  const int b = px / hw, rem = px - b * hw, h = rem / width, w = rem - h * width;
  const int nbins = pooled * pooled;
  for (int i = tid; i < nbins * CP; i += 256) accum[i] = 0.f;
  const float inv_bins = 1.f / (float)nbins;
  // (chunks of 256 RoIs in RoI order, from the first to the last RoI of this pixel's image)
  const unsigned last1 = range[2 * b + 1];
  const int r_first = last1 ? (int)~range[2 * b] : 0, r_end = (int)last1;
  for (int r0 = r_first; r0 < r_end; r0 += 256) {
    // ---- phase 1: RoI r0 + tid -> the bins of that RoI that contain this pixel
    const int r = r0 + tid;
    int nh = 0, ph_lo = 0, ph_hi = -1, pw_lo = 0, pw_hi = -1;
    const int* e = edges + min(r, num_rois - 1);
    const long ns = num_rois;
    if (r < r_end && e[4 * pooled * ns] == b) {
      // bins are intervals with non-decreasing edges: the ph whose [hstart, hend) contains h form a contiguous run
      ph_lo = pooled; pw_lo = pooled;
      if constexpr (POOLED > 0) {
        int ev[4 * POOLED];
#pragma unroll
        for (int k = 0; k < 4 * POOLED; ++k) ev[k] = e[k * ns];
#pragma unroll
        for (int k = 0; k < POOLED; ++k) {
          if (ev[k] <= h && h < ev[POOLED + k]) { ph_lo = min(ph_lo, k); ph_hi = k; }
          if (ev[2 * POOLED + k] <= w && w < ev[3 * POOLED + k]) { pw_lo = min(pw_lo, k); pw_hi = k; }
        }
      } else {
        for (int k = 0; k < pooled; ++k) {
          if (e[k * ns] <= h && h < e[(pooled + k) * ns]) { ph_lo = min(ph_lo, k); ph_hi = k; }
          if (e[(2 * pooled + k) * ns] <= w && w < e[(3 * pooled + k) * ns]) { pw_lo = min(pw_lo, k); pw_hi = k; }
        }
      }
      if (ph_hi >= 0 && pw_hi >= 0) nh = (ph_hi - ph_lo + 1) * (pw_hi - pw_lo + 1);
    }
    // ordered compaction: exclusive prefix of nh over the 256 RoIs of the chunk (wave scan + 4 wave totals)
    int incl = nh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += wave_cnt[k];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    int pos = base + incl - nh;
    if (nh > 0 && total <= kMaxHits) {
      for (int ph = ph_lo; ph <= ph_hi; ++ph)
        for (int pw = pw_lo; pw <= pw_hi; ++pw) {
          const int area = (e[(pooled + ph) * ns] - e[ph * ns]) * (e[(3 * pooled + pw) * ns] - e[(2 * pooled + pw) * ns]);
          hit_r[pos] = r; hit_bin[pos] = (short)(ph * pooled + pw); hit_w[pos] = inv_bins / (float)area;
          ++pos;
        }
    }
    __syncthreads();
    // ---- phase 2: wave v owns the bins = v (mod 4); it walks the list in order (= RoI order), lanes = classes
    if (total <= kMaxHits) {
      // (eight gradient rows requested before the first add: the walk is a chain of dependent global loads otherwise -- 30 hits
      //  per pixel at the training shape, one memory round trip each; the adds stay in list order)
      constexpr int NF = 8;
      for (int i0 = 0; i0 < total; i0 += NF) {
        float gv[NF], ww[NF];
        int bn[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
          const int i = min(i0 + u, total - 1);
          const int bin = hit_bin[i];
          const bool mine = i0 + u < total && (bin & 3) == wave;
          bn[u] = mine ? bin : -1;
          ww[u] = hit_w[i];
          gv[u] = (mine && lane < output_dim) ? gvote[(long)hit_r[i] * output_dim + lane] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NF; ++u)
          if (bn[u] >= 0 && lane < CP) accum[bn[u] * CP + lane] += gv[u] * ww[u];
      }
    } else {
      // (pathological chunk: more than 8 bins per RoI on average -- every RoI handled by one thread-serial pass, still in order)
      if (tid < 64) {
        for (int rr = r0; rr < min(r0 + 256, r_end); ++rr) {
          const int* ee = edges + rr;
          const long ns = num_rois;
          if (ee[4 * pooled * ns] != b) continue;
          for (int ph = 0; ph < pooled; ++ph) {
            if (!(ee[ph * ns] <= h && h < ee[(pooled + ph) * ns])) continue;
            for (int pw = 0; pw < pooled; ++pw) {
              if (!(ee[(2 * pooled + pw) * ns] <= w && w < ee[(3 * pooled + pw) * ns])) continue;
              const int area = (ee[(pooled + ph) * ns] - ee[ph * ns]) * (ee[(3 * pooled + pw) * ns] - ee[(2 * pooled + pw) * ns]);
              if (lane < CP) accum[(ph * pooled + pw) * CP + lane] += (lane < output_dim ? gvote[(long)rr * output_dim + lane] : 0.f) * (inv_bins / (float)area);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  float* dst = gmap + (long)px * pixel_stride;
  for (int i = tid; i < nbins * CP; i += 256) dst[i] = accum[i];
}

template <int TPX, int NTW, int NLOAD, bool PIN, int NSTAGE = 3, bool RPN = false>
int launch_head(const HeadGeom& g, hipStream_t stream) {
  constexpr size_t lds = (size_t)NSTAGE * (TPX * 16 + 4 * NTW * 16) * kBK * sizeof(float);
  static_assert(lds <= 160 * 1024, "three stages must fit the CU's LDS");
  static DttDeviceOnce once;
  bool& raised_here = once.here();
  if (lds > 64 * 1024 && !raised_here) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_gemm_kernel<TPX, NTW, NLOAD, PIN, NSTAGE, RPN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DTT_REQUIRE(e == hipSuccess, "head_gemm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    raised_here = true;
  }
  const char* tag = RPN ? "rpn_head_gemm" : "head_gemm";   // (the RPN launch carries its own timing tag)
  dtt_prof_begin(tag, stream);
  hipLaunchKernelGGL((head_gemm_kernel<TPX, NTW, NLOAD, PIN, NSTAGE, RPN>), dim3(g.n_groups * g.strips), dim3((4 + NLOAD) * 64), lds, stream, g);
  dtt_prof_end(tag, stream);
  DTT_CHECK_LAUNCH("head_gemm");
  return 1;
}

// tiles per pass for a group of `per_group` tiles: full passes of 4 * ntw tiles, then the remainder; a caller asking for
// more passes gets the tail split further (the last pass is what is left to store when the MFMAs end -- every earlier
// pass drains underneath its successor -- but each pass re-streams the pixel strip)
void plan_passes(HeadGeom& g, int per_group, int ntw, int passes) {
  const int cap = 4 * ntw;
  int n = 0, left = per_group;
  while (left > 0 && n < kMaxPasses - 1) {
    int len = min(cap, left);
    const int full_left = dtt_cdiv(left - len, cap);             // passes the greedy plan still needs after this one
    if (passes > 0 && n + 1 + full_left < passes && len > 4 && left > 4) {
      // spread what is left over the passes still wanted, in multiples of the 4 waves
      const int want = passes - n;
      len = min(cap, max(4, dtt_cdiv(dtt_cdiv(left, want), 4) * 4));
    }
    g.pass_len[n++] = len;
    left -= len;
  }
  if (left > 0) g.pass_len[n++] = left;
  g.passes = n;
}

}  // namespace

#ifdef DTT_HEAD_STAMP
extern "C" int dtt_head_stamps_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_head_stamps), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

static int head_gemm_launch(HeadGeom g, const float* x, long ldx, int M, int K, const float* w, const float* bias, int n_rows,
                            float* out, long ldc, int n_store, int passes, hipStream_t stream) {
  DTT_REQUIRE(x && w && bias && out, "head_gemm: null pointer");
  DTT_REQUIRE(M > 0 && K > 0 && K % kBK == 0, "head_gemm: K (%d) must be a positive multiple of %d", K, kBK);
  DTT_REQUIRE(n_rows > 0 && n_rows % 16 == 0, "head_gemm: weight rows (%d) must be padded to a multiple of 16", n_rows);
  DTT_REQUIRE(n_store > 0 && n_store <= n_rows && n_store % 4 == 0 && ldc >= n_store && ldc % 4 == 0 && ldx % 4 == 0 && ldx >= K,
              "head_gemm: bad leading dimensions / n_store");
  DTT_REQUIRE((((size_t)x | (size_t)w | (size_t)out | (size_t)bias) & 15) == 0, "head_gemm: pointers must be 16-byte aligned");
  g.x = x; g.ldx = ldx; g.w = w; g.bias = bias; g.out = out; g.ldc = ldc;
  g.M = M; g.K = K; g.n_store = n_store; g.nt_total = n_rows / 16;
  static const int ablate = getenv("DTT_HEAD_ABLATE") ? atoi(getenv("DTT_HEAD_ABLATE")) : 0;
  g.ablate = ablate;
  const int ncu = dtt_device_cus();
  static const int nload = getenv("DTT_HEAD_NLOAD") ? atoi(getenv("DTT_HEAD_NLOAD")) : 4;   // developer A/B switches
  if (g.nt_total > 16) {   // more channel tiles than the narrow configuration's single pass holds
    // wide heads (31*49 classes [+ 4*49 box deltas]): pixel strips x channel groups, one workgroup per CU
    static const int ntw_env = getenv("DTT_HEAD_NTW") ? atoi(getenv("DTT_HEAD_NTW")) : 4;
    static const int pin_env = getenv("DTT_HEAD_PIN") ? atoi(getenv("DTT_HEAD_PIN")) : 0;
    const int tpx = 10, ntw = 4;
    g.strips = dtt_cdiv(M, tpx * 16);
    g.n_groups = max(1, min(ncu / max(g.strips, 1), g.nt_total));
    while (dtt_cdiv(g.nt_total, g.n_groups) > kMaxPasses * 4 * ntw) ++g.n_groups;
    plan_passes(g, dtt_cdiv(g.nt_total, g.n_groups), ntw, passes);
    (void)ntw_env; (void)pin_env;
    DTT_REQUIRE(g.epilogue == 0, "rpn_head_gemm: %d packed output rows exceed the narrow configurations (<= 256) the RPN epilogue is built for", g.nt_total * 16);
    DTT_REQUIRE((unsigned long long)M * (unsigned long long)ldc * 4ull < (1ull << 32), "head_gemm: output of %d x %ld floats exceeds the 32-bit store offsets", M, ldc);
    return nload == 2 ? launch_head<10, 4, 2, false>(g, stream) : launch_head<10, 4, 4, false>(g, stream);
  }
  static const int narrow_cfg = getenv("DTT_HEAD_NARROW") ? atoi(getenv("DTT_HEAD_NARROW")) : 61;   // developer A/B switch (24 = the small-M form)
  if ((narrow_cfg == 61 || narrow_cfg == 51) && M >= 2048) {
    // narrow heads over many pixels (corr_bbox_net: 5092 x 1056 x 196): 96-pixel strips x groups of 4 channel tiles, one tile per
    // compute wave, one pass -- 54 strips x 4 groups = 216 workgroups at the 600 px shape: 34 us standalone (32-pixel strips with
    // all 13 tiles in one workgroup: 160 workgroups, 41 us).  80-pixel strips (256 workgroups, 51) are 29 us standalone, but a grid
    // that fills every CU leaves no room for the NMS sweep's four 1024-thread workgroups that run beside it (sweep 25 -> 50 us).
    // very narrow heads (the RPN's 72 channels = 5 tiles): the waves of a workgroup split CHANNEL tiles, so the per-wave MFMA
    // chain (pixel tiles x K) is what bounds the launch -- half-length strips, twice the workgroups (two per CU)
    const int tpx = g.nt_total <= 6 ? 3 : narrow_cfg == 51 ? 5 : 6;
    g.strips = dtt_cdiv(M, tpx * 16);
    g.n_groups = dtt_cdiv(g.nt_total, 4);
    plan_passes(g, dtt_cdiv(g.nt_total, g.n_groups), 1, 1);
    DTT_REQUIRE((unsigned long long)M * (unsigned long long)ldc * 4ull < (1ull << 32), "head_gemm: output of %d x %ld floats exceeds the 32-bit store offsets", M, ldc);
    if (g.epilogue == 1) {   // the RPN's heads: the epilogue with the pairwise softmax is compiled into these instantiations only
      if (tpx == 3) return launch_head<3, 1, 2, false, 4, true>(g, stream);
      if (tpx == 6) return launch_head<6, 1, 2, false, 4, true>(g, stream);
      return launch_head<5, 1, 2, false, 4, true>(g, stream);
    }
    if (tpx == 3) return launch_head<3, 1, 2, false, 4>(g, stream);
    if (tpx == 6) return launch_head<6, 1, 2, false, 4>(g, stream);
    return launch_head<5, 1, 2, false, 4>(g, stream);
  }
  // narrow heads (4*49 box deltas alone): 32-pixel strips, every tile of the row in one pass
  constexpr int TPX = 2, NTW = 4;
  DTT_REQUIRE(g.nt_total <= 4 * NTW, "head_gemm: %d channel tiles not covered by the narrow configuration", g.nt_total);
  g.strips = dtt_cdiv(M, TPX * 16);
  g.n_groups = 1;
  plan_passes(g, g.nt_total, NTW, 1);
  DTT_REQUIRE((unsigned long long)M * (unsigned long long)ldc * 4ull < (1ull << 32), "head_gemm: output of %d x %ld floats exceeds the 32-bit store offsets", M, ldc);
  if (g.epilogue == 1) return launch_head<TPX, NTW, 1, false, 3, true>(g, stream);
  return launch_head<TPX, NTW, 1, false>(g, stream);
}

// out[m][n] = sum_k x[m][k] * w[n][k] + bias[n]  for n < n_store  (the 1x1 convolution of rfcn.py:49-53 over
// channels-last pixel rows, channels emitted in the order of w's rows).
extern "C" int dtt_head_gemm(const float* x, long ldx, int M, int K, const float* w, const float* bias, int n_rows,
                             float* out, long ldc, int n_store, int passes, void* stream_) {
  HeadGeom g;
  g.epilogue = 0; g.A = 0; g.hw = 1; g.hw_magic = 0; g.out2 = nullptr;
  return head_gemm_launch(g, x, ldx, M, K, w, bias, n_rows, out, ldc, n_store, passes, static_cast<hipStream_t>(stream_));
}

// The RPN's two 1x1 heads + the pairwise softmax in ONE launch (rpn.py:63-71: RPN_cls_score, reshape(2) -> softmax ->
// reshape(2A), RPN_bbox_pred).  x: (batch * hw, K) channels-last rows of relu(RPN_Conv(.)).  w: (n_rows, K), rows emitted in
// the order [bg_0, fg_0, bg_1, fg_1, ..., bg_{A-1}, fg_{A-1}, box deltas 0 .. 4A-1, zero padding to a multiple of 16]
// (bg_a = RPN_cls_score channel a, fg_a = channel A + a: the pair the reference's softmax normalises); bias alike.
// cls_prob: (batch, 2A, h, w) and bbox_pred: (batch, 4A, h, w), NCHW -- what dtt_proposal_select_sort / _decode_nms read.
extern "C" int dtt_rpn_head_gemm(const float* x, long ldx, int batch, int hw, int K, const float* w, const float* bias, int n_rows,
                                 int num_anchors, float* cls_prob, float* bbox_pred, void* stream_) {
  DTT_REQUIRE(cls_prob && bbox_pred, "rpn_head_gemm: null pointer");
  DTT_REQUIRE(batch > 0 && hw > 0 && num_anchors > 0 && (long)batch * hw < (1l << 31) && n_rows >= 6 * num_anchors && num_anchors % 2 == 0,
              "rpn_head_gemm: bad shape (an even number of anchors, weight rows >= 6 * anchors)");
  HeadGeom g;
  g.epilogue = 1; g.A = num_anchors; g.hw = hw; g.hw_magic = 0xffffffffu / (unsigned)hw + 1u; g.out2 = bbox_pred;
  DTT_REQUIRE(hw > 1, "rpn_head_gemm: a map of one pixel is not supported");
  // the epilogue splits a row index into (image, pixel) with __umulhi(row, 2^32 / hw + 1): exact while row * hw < 2^32
  DTT_REQUIRE((unsigned long long)batch * hw * hw < (1ull << 32), "rpn_head_gemm: batch * hw^2 = %llu exceeds the 32-bit range of the "
              "row -> (image, pixel) split", (unsigned long long)batch * hw * hw);
  const int n_store = 6 * num_anchors;
  return head_gemm_launch(g, x, ldx, batch * hw, K, w, bias, n_rows, cls_prob, (long)((n_store + 3) / 4 * 4), (n_store + 3) / 4 * 4, 1,
                          static_cast<hipStream_t>(stream_));
}

// Backward of the RPN epilogue (training graph, rpn.py:63-71): the gradients of the probabilities and box deltas, which arrive as
// the reference's NCHW tensors, become the rows of the packed GEMM's output gradient -- columns [bg_0, fg_0, bg_1, fg_1, ...,
// box deltas 0 .. 4A-1, zeros up to `ld`] -- with the adjoint of the pairwise softmax applied on the way:
//   d/ds_bg = p_bg * (g_bg - (g_bg p_bg + g_fg p_fg)),   d/ds_fg = p_fg * (g_fg - (g_bg p_bg + g_fg p_fg)).
// One thread per pixel row: plane reads are coalesced across the threads of a wave, a thread writes its row as 16-byte pieces.
namespace {
__global__ __launch_bounds__(256) void rpn_head_grad_rows_kernel(const float* __restrict__ g_prob, const float* __restrict__ g_bbox,
                                                                 const float* __restrict__ prob, int batch, int hw, int A,
                                                                 float* __restrict__ rows, long ld, int logits) {
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= (long)batch * hw) return;
  const int b = (int)(m / hw), p = (int)(m - (long)b * hw);
  float* out = rows + m * ld;
  const float* pp = prob + (long)b * 2 * A * hw + p;
  const float* gp = g_prob ? g_prob + (long)b * 2 * A * hw + p : nullptr;
  const float* gb = g_bbox ? g_bbox + (long)b * 4 * A * hw + p : nullptr;
  for (int a = 0; a < A; a += 2) {      // two (background, foreground) pairs = one 16-byte piece (A is even)
    float4 o;
    float* ov = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float pb = pp[(long)(a + u) * hw], pf = pp[(long)(A + a + u) * hw];
      const float g0 = gp ? gp[(long)(a + u) * hw] : 0.f, g1 = gp ? gp[(long)(A + a + u) * hw] : 0.f;
      const float dot = g0 * pb + g1 * pf;
      ov[2 * u] = logits ? g0 : pb * (g0 - dot);          // (logits: g is the gradient with respect to the scores already)
      ov[2 * u + 1] = logits ? g1 : pf * (g1 - dot);
    }
    *reinterpret_cast<float4*>(out + 2 * a) = o;
  }
  for (int j = 0; j < 4 * A; j += 4) {
    float4 o;
    o.x = gb ? gb[(long)j * hw] : 0.f; o.y = gb ? gb[(long)(j + 1) * hw] : 0.f;
    o.z = gb ? gb[(long)(j + 2) * hw] : 0.f; o.w = gb ? gb[(long)(j + 3) * hw] : 0.f;
    *reinterpret_cast<float4*>(out + 2 * A + j) = o;
  }
  for (long j = 6 * A; j < ld; j += 4) *reinterpret_cast<float4*>(out + j) = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

extern "C" int dtt_rpn_head_grad_rows(const float* grad_cls_prob, const float* grad_bbox_pred, const float* cls_prob, int batch, int hw,
                                      int num_anchors, float* rows, long ld, int cls_grad_is_logits, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(cls_prob && rows && batch > 0 && hw > 0 && num_anchors > 0 && num_anchors % 2 == 0, "rpn_head_grad_rows: bad arguments");
  DTT_REQUIRE(ld >= 6 * num_anchors && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0,
              "rpn_head_grad_rows: rows need a 16-byte aligned base and a row length (%ld) that is a multiple of 4 >= %d", ld, 6 * num_anchors);
  const long M = (long)batch * hw;
  hipLaunchKernelGGL(rpn_head_grad_rows_kernel, dim3(dtt_cdiv(M, 256)), dim3(256), 0, stream, grad_cls_prob, grad_bbox_pred, cls_prob, batch,
                     hw, num_anchors, rows, ld, cls_grad_is_logits);
  DTT_CHECK_LAUNCH("rpn_head_grad_rows");
  return 1;
}

// Backward of dtt_psroi_pm_forward's vote: grad_map (batch*height*width pixels, pixel_stride floats apart; this call writes
// floats [bin*cp + c] for bin < pooled^2, c < cp of EVERY pixel -- zeros where no RoI reaches, so no pre-zeroing) from
// grad_vote (num_rois, output_dim).  edges: caller-owned scratch of num_rois * (4 * pooled + 1) + 2 * batch_size ints.
// (round 6: the exported dtt_psroi_pm_backward lives in psroi_bwd.hip -- one wave per pixel, all heads in one launch; this is the
//  one-workgroup-per-pixel kernel of rounds 4 - 5, reached with DTT_PSROI_BWD_OLD=1 for the A/B and for pooled sizes other than 7)
__attribute__((visibility("hidden"))) int dtt_psroi_pm_backward_old(const float* grad_vote, const float* rois, int num_rois, int batch_size,
                                                                    int height, int width, int pooled, float spatial_scale, int output_dim,
                                                                    int cp, long pixel_stride, float* grad_map, int* edges, hipStream_t stream) {
  DTT_REQUIRE(batch_size > 0 && height > 0 && width > 0 && pooled > 0 && output_dim > 0 && num_rois >= 0, "psroi_pm backward: bad shape");
  DTT_REQUIRE(cp >= output_dim && (long)pooled * pooled * cp <= pixel_stride, "psroi_pm backward: %d bins x %d do not fit the pixel stride %ld",
              pooled * pooled, cp, pixel_stride);
  DTT_REQUIRE(grad_map && edges && (num_rois == 0 || (grad_vote && rois)), "psroi_pm backward: null pointer");
  DTT_REQUIRE(cp == 4 || cp == 32, "psroi_pm backward: classes-per-bin padding %d not instantiated (4 or 32)", cp);
  const size_t lds = (size_t)pooled * pooled * cp * sizeof(float);
  DTT_REQUIRE(lds <= 32 * 1024, "psroi_pm backward: pooled size too large");
  // per-image RoI runs behind the bin edges: 2 words per image, zeroed here, filled by the edges kernel
  unsigned* range = reinterpret_cast<unsigned*>(edges + (long)(num_rois > 0 ? num_rois : 0) * (4 * pooled + 1));
  DTT_REQUIRE(hipMemsetAsync(range, 0, sizeof(unsigned) * 2 * batch_size, stream) == hipSuccess, "psroi_pm backward: memset failed");
  dtt_prof_begin("psroi_pm_bwd", stream);   // (event tag: edges + the map-stationary gradient kernel)
  if (num_rois > 0) {
    hipLaunchKernelGGL(psroi_pm_edges_kernel, dim3(dtt_cdiv(num_rois, 256)), dim3(256), 0, stream, rois, num_rois, spatial_scale, pooled,
                       height, width, batch_size, edges, range, grad_vote, output_dim);
    DTT_CHECK_LAUNCH("psroi_pm_edges");
  }
  const int npx = batch_size * height * width;
#define DTT_PMB_LAUNCH(CPV, PV)                                                                                                  \
  hipLaunchKernelGGL((psroi_pm_bwd_kernel<CPV, PV>), dim3(npx), dim3(256), lds, stream, grad_vote, edges, num_rois, output_dim, pooled, \
                     height, width, pixel_stride, grad_map, range)
  if (cp == 32) { if (pooled == 7) DTT_PMB_LAUNCH(32, 7); else DTT_PMB_LAUNCH(32, 0); }
  else { if (pooled == 7) DTT_PMB_LAUNCH(4, 7); else DTT_PMB_LAUNCH(4, 0); }
#undef DTT_PMB_LAUNCH
  dtt_prof_end("psroi_pm_bwd", stream);
  DTT_CHECK_LAUNCH("psroi_pm_bwd");
  return 1;
}

// Position-sensitive pooling + vote over a position-major map (see the header comment).  map: (batch, H, W) pixels of
// `pixel_stride` floats; this call reads floats [bin*cp + c] of a pixel for bin < pooled^2, c < output_dim (callers point
// `map` at the first float of the head they pool: class scores or box deltas).  vote_out (num_rois, output_dim);
// pooled_out (num_rois, output_dim, pooled, pooled) or NULL.
extern "C" int dtt_psroi_pm_forward(const float* map, long pixel_stride, int cp, int batch_size, int num_rois, int height,
                                    int width, int pooled, const float* rois, float spatial_scale, int output_dim,
                                    float* vote_out, float* pooled_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(batch_size > 0 && height > 0 && width > 0 && pooled > 0 && output_dim > 0 && num_rois >= 0, "psroi_pm: bad shape");
  DTT_REQUIRE(cp >= output_dim && (long)pooled * pooled * cp <= pixel_stride, "psroi_pm: %d bins x %d do not fit the pixel stride %ld",
              pooled * pooled, cp, pixel_stride);
  if (num_rois == 0) return 1;
  DTT_REQUIRE(map && rois && vote_out, "psroi_pm: null pointer");
  DTT_REQUIRE(((size_t)map & 15) == 0 && pixel_stride % 4 == 0, "psroi_pm: the map must be 16-byte aligned with a pixel stride that is a multiple of 4 floats");
  const size_t lds = (size_t)pooled * pooled * cp * sizeof(float);
  DTT_REQUIRE(lds <= 64 * 1024, "psroi_pm: pooled size too large");
  dtt_prof_begin("psroi_pm", stream);
#define DTT_PM_LAUNCH(CPV, LPBV, NWV, PV)                                                                                   \
  hipLaunchKernelGGL((psroi_pm_kernel<CPV, LPBV, NWV, PV>), dim3(num_rois), dim3(NWV * 64), lds, stream, map, pixel_stride, \
                     height, width, rois, spatial_scale, pooled, output_dim, vote_out, pooled_out, batch_size)
  if (cp == 32) {
    if (pooled == 7) DTT_PM_LAUNCH(32, 4, 4, 7); else DTT_PM_LAUNCH(32, 4, 4, 0);
  } else if (cp == 4) {
    if (pooled == 7) DTT_PM_LAUNCH(4, 1, 1, 7); else DTT_PM_LAUNCH(4, 1, 1, 0);
  } else {
    dtt_set_error("psroi_pm: classes-per-bin padding %d not instantiated (4 or 32)", cp);
    return 0;
  }
#undef DTT_PM_LAUNCH
  dtt_prof_end("psroi_pm", stream);
  DTT_CHECK_LAUNCH("psroi_pm");
  return 1;
}

// Class scores + box deltas of every RoI in ONE launch, softmax folded in (rfcn.py:133-140 at inference): map as for
// dtt_psroi_pm_forward; the class head at float 0 of a pixel (32 slots per bin, n_cls <= 32 used), the box head at float loc_offset
// (4 per bin, n_loc <= 4).  cls_prob (num_rois, n_cls), loc_vote (num_rois, n_loc); cls_vote (num_rois, n_cls) or NULL.
extern "C" int dtt_psroi_pm_det_forward(const float* map, long pixel_stride, int loc_offset, int batch_size, int num_rois, int height, int width,
                                        int pooled, const float* rois, float spatial_scale, int n_cls, int n_loc, float* cls_vote,
                                        float* cls_prob, float* loc_vote, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(batch_size > 0 && height > 0 && width > 0 && pooled > 0 && num_rois >= 0, "psroi_pm_det: bad shape");
  DTT_REQUIRE(n_cls >= 1 && n_cls <= 32 && n_loc >= 1 && n_loc <= 4, "psroi_pm_det: %d classes / %d box outputs not covered (<= 32 / <= 4)", n_cls, n_loc);
  DTT_REQUIRE(loc_offset >= pooled * pooled * 32 && loc_offset % 4 == 0 && (long)loc_offset + pooled * pooled * 4 <= pixel_stride,
              "psroi_pm_det: the box head (float %d) must lie behind the %d class bins inside the pixel stride %ld", loc_offset, pooled * pooled, pixel_stride);
  if (num_rois == 0) return 1;
  DTT_REQUIRE(map && rois && cls_prob && loc_vote, "psroi_pm_det: null pointer");
  DTT_REQUIRE(((size_t)map & 15) == 0 && pixel_stride % 4 == 0, "psroi_pm_det: the map must be 16-byte aligned with a pixel stride that is a multiple of 4 floats");
  const size_t lds = (size_t)pooled * pooled * 36 * sizeof(float);
  DTT_REQUIRE(lds <= 64 * 1024, "psroi_pm_det: pooled size too large");
  dtt_prof_begin("psroi_pm", stream);
  if (pooled == 7)
    hipLaunchKernelGGL((psroi_pm_det_kernel<7>), dim3(num_rois), dim3(320), lds, stream, map, pixel_stride, loc_offset, height, width, rois,
                       spatial_scale, pooled, n_cls, n_loc, cls_vote, cls_prob, loc_vote, batch_size);
  else
    hipLaunchKernelGGL((psroi_pm_det_kernel<0>), dim3(num_rois), dim3(320), lds, stream, map, pixel_stride, loc_offset, height, width, rois,
                       spatial_scale, pooled, n_cls, n_loc, cls_vote, cls_prob, loc_vote, batch_size);
  dtt_prof_end("psroi_pm", stream);
  DTT_CHECK_LAUNCH("psroi_pm_det");
  return 1;
}

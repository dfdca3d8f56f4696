// Cross-frame correlation on channels-last feature maps (gfx950): the forward of Correlation_forward
// (correlation/src/correlation_cuda_kernel.cu:34-106) for kernel_size 1, stride1 == stride2, max_displacement / stride <= 8
// (or 12 / 16: four 17 x 17 sub-windows as virtual images of the same launch) -- the three correlations of D&T
// (rfcn.py:58-60, 170-172) -- reading the trunk's channels-last maps directly.
//
// The reference first repacks both NCHW maps to NHWC (`channels_first`, .cu:10-32) and then walks them with one 32-thread
// block per output pixel.  The channels-last trunk of this repo already produces NHWC, so this kernel consumes it as is
// (no layout transposes in front of the op) and runs the banded product  out[p, q] = 1/C * sum_c f1[p, c] * f2[q, c],
// |q - p| <= R, on the matrix cores with the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: an fma chain, bit for bit).
//
//   * stride s > 1 (conv3: s = 2) touches only the pixels of the s-lattice, so it IS the stride-1 problem on the
//     sub-sampled map: the source is addressed with pixel strides (s * C, s * W * C), nothing is copied.
//   * One workgroup = one 8 x 12 pixel tile x one channel slice; 12 waves = 6 blocks of 4 x 4 frame-t pixels (MFMA
//     columns) x 2 halves of each 16-channel chunk.  A wave multiplies its pixel block against the (NBR x NBR) 4 x 4
//     blocks of frame-(t+tau) pixels of its window (MFMA rows): NBR^2 accumulators, both operands K-contiguous, so one
//     8-byte LDS read per operand feeds two MFMAs.
//   * Staging global -> LDS by LDS-DMA in the scalar-base form (SALU + VMEM only), three stages, one barrier per chunk.
//     The 16-byte pieces of a pixel's 64-byte chunk row are XOR-swizzled on the source side so that the operand reads are
//     bank-conflict free.  Zero padding is applied on the output side (out-of-image pixels are staged from a clamped
//     address and their products discarded).
//   * The two channel halves of a workgroup meet in LDS; the channel slices of a tile meet through one slab each in the
//     workspace: a slice publishes its slab (release), takes a ticket, and the LAST arriver of the tile (acquire) sums
//     the slabs in slice order -- deterministic -- divides by C and writes the window entries in the caller's layout
//     (NCHW planes or position-major rows).  No second kernel, no split-K partials beyond one slab per slice.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include "common.h"
#include "corr_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kNWaves = 12, kNThreads = kNWaves * 64;
constexpr int kKC = 16;              // channels per stage
constexpr int kTY = 8, kTX = 12;     // output tile (pixels): 2 x 3 blocks of 4 x 4

struct NGeom {
  const float* f1; const float* f2;  // frame t / t+tau, channels-last
  long sy, sx, sb;                   // floats between vertically / horizontally adjacent lattice pixels, between images
  int C, H, W;                       // channels, lattice size (pixels the correlation can touch)
  int oh, ow, origin;                // output size; output (y, x) <-> lattice pixel (origin + y, origin + x)
  int R, Rfull, Dfull;               // sub-window radius, full window
  int nquad, qyv[4], qxv[4];         // 8 < R <= 16: four (2*8+1)^2 sub-windows with these centre shifts, run as four "virtual images" of ONE launch
  int tiles_x, tiles_y, ksplit, c_per_split, batch;
  float* out; long out_sb, out_sc, out_sp;   // element (n, d, y, x) at out[n*sb + d*sc + (y*ow + x)*sp]
  float* slabs; int* tickets;
  int trace_slot;
  int ablate;   // developer timing experiments (DTT_CORR_NHWC_ABLATE): 1 no DMA, 2 no MFMA, 4 no epilogue
};

__device__ __forceinline__ void dma16n(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(lds_addr), "v"(voff), "s"(sbase)
               : "memory", "m0");
}
__device__ __forceinline__ const char* uptr(const char* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  return (const char*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

#ifdef DTT_WG_TRACE   // developer build (tools/build_variant.sh): where and when every workgroup of the conv5 launches ran
__device__ unsigned long long dtt_nhwc_trace[16 * 256 * 4];
#define NHWC_TRACE(slot, v) do { if (g.trace_slot >= 0 && threadIdx.x == 0 && blockIdx.x < 256) \
    dtt_nhwc_trace[(g.trace_slot * 256 + blockIdx.x) * 4 + (slot)] = (v); } while (0)
__device__ __forceinline__ unsigned long long nhwc_hw_id() {
  unsigned id, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((unsigned long long)(xcc & 15) << 16) | (((id >> 13) & 7) << 8) | ((id >> 8) & 15);
}
#else
#define NHWC_TRACE(slot, v) do {} while (0)
#endif

template <int NBR>
struct NCfg {
  static constexpr int R = 2 * (NBR - 1);            // window radius covered: 8 (NBR 5) or 4 (NBR 3)
  static constexpr int HR = kTY + 2 * R, HC = kTX + 2 * R;   // halo rows / cols
  static constexpr int HPX = HR * HC, PPX = kTY * kTX, NPX = HPX + PPX;
  static constexpr int NI = (NPX / 16 + kNWaves - 1) / kNWaves;   // DMA instructions per wave per chunk (16 pixels each)
  static constexpr int NIMIN = (NPX / 16) / kNWaves;              // what every wave issues at least: the counted wait
  static constexpr int STAGE = NPX * kKC;            // floats
  static constexpr int NB = NBR * NBR;
  static constexpr size_t LDS = 3ul * STAGE * sizeof(float);
  static_assert(NPX % 16 == 0 && HC % 4 == 0, "pixel rows of the LDS image come in whole 16-pixel DMA instructions");
};

template <int NBR>
__global__ __launch_bounds__(kNThreads) void corr_nhwc_kernel(NGeom g) {
  using K = NCfg<NBR>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int ticket_s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = g.tiles_x * g.tiles_y;
#ifdef DTT_WG_TRACE
  NHWC_TRACE(0, wall_clock64());
  NHWC_TRACE(1, nhwc_hw_id());
#endif
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  // The tiles of one (image, channel slice) sit side by side: an XCD's contiguous run of items shares that slice's
  // halo pixels through its L2 (30 tiles x 4 slices x 2 images = one (image, slice) per XCD at the 600 px shapes).  The
  // slices of a tile then run on different XCDs; their slabs meet in memory (write-through stores, agent-scope acquire).
  int ks, tile, n;
  if (g.ablate & 64) { ks = item % g.ksplit; tile = (item / g.ksplit) % ntiles; n = item / (g.ksplit * ntiles); }
  else { tile = item % ntiles; ks = (item / ntiles) % g.ksplit; n = item / (g.ksplit * ntiles); }
  // (n counts virtual images: sub-window q of image n % batch; tickets / slabs are per virtual image)
  const int quad = n / g.batch, n_img = n - quad * g.batch;
  const int qy = g.qyv[quad], qx = g.qxv[quad];
  const int ty0 = (tile / g.tiles_x) * kTY, tx0 = (tile % g.tiles_x) * kTX;
  const int c_begin = ks * g.c_per_split, c_end = min(g.C, c_begin + g.c_per_split);
  const int nch = (c_end - c_begin) / kKC;
  const int pb = wave % 6, kh = wave / 6, by = pb / 3, bx = pb % 3;

  // ---- DMA plan: pixel (instr * 16 + lane / 4), 16-byte piece (lane % 4) ^ swizzle; the source pixel is clamped into
  // the image (its products are discarded on the output side)
  unsigned voff[K::NI];
  bool from_f1[K::NI];
#pragma unroll
  for (int i = 0; i < K::NI; ++i) {
    const int instr = i * kNWaves + wave;
    const int px = min(instr * 16 + (lane >> 2), K::NPX - 1);
    int y, x, row;
    if (px < K::HPX) {
      row = px / K::HC;
      y = ty0 - K::R + row + qy + g.origin;
      x = tx0 - K::R + px % K::HC + qx + g.origin;
    } else {
      row = (px - K::HPX) / kTX;
      y = ty0 + row + g.origin;
      x = tx0 + (px - K::HPX) % kTX + g.origin;
    }
    const int piece = (lane & 3) ^ (row & 3);   // swizzle key = pixel row: the 4 rows of a block land in different bank groups
    y = min(max(y, 0), g.H - 1);
    x = min(max(x, 0), g.W - 1);
    voff[i] = (unsigned)(((long)y * g.sy + (long)x * g.sx) * 4 + piece * 16);
    from_f1[i] = instr * 16 >= K::HPX;   // wave-uniform: the frame-t tile starts on an instruction boundary
  }
  static_assert(K::HPX % 16 == 0, "frame boundary on a DMA instruction boundary");
  const char* b1 = reinterpret_cast<const char*>(g.f1 + (long)n_img * g.sb + c_begin);
  const char* b2 = reinterpret_cast<const char*>(g.f2 + (long)n_img * g.sb + c_begin);
  const unsigned lds0 = (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)lds;
  auto issue = [&](int ci) {
    const unsigned st = lds0 + (unsigned)((ci % 3) * K::STAGE * 4);
    const char* s1 = uptr(b1 + (long)ci * kKC * 4);
    const char* s2 = uptr(b2 + (long)ci * kKC * 4);
#pragma unroll
    for (int i = 0; i < K::NI; ++i) {
      const int instr = i * kNWaves + wave;
      if (instr * 16 < K::NPX && !(g.ablate & 1)) dma16n(from_f1[i] ? s1 : s2, voff[i], st + instr * 1024);
    }
  };

  // ---- operand addresses (floats inside a stage).  A = frame-(t+tau) block rows, B = frame-t block; lane (i = lane % 16,
  // g = lane / 16) reads channels {kh*8 + 2g, +1} of pixel i: one 8-byte read, two MFMA k-steps.
  const int li = lane & 15, lg = lane >> 4, iy = li >> 2, ix = li & 3;
  // 8-byte slot (kh*4 + g) of the pixel's 64-byte chunk row, XOR 2 * (pixel row % 4): with the pixel column % 4 selecting
  // the 64-byte quarter of the 256-byte bank line, the 32 lanes of a ds_read_b64 group hit 32 different 8-byte slots
  const int sl = ((kh * 4 + lg) ^ (iy << 1)) << 1;           // every row this lane reads has row % 4 == iy
  const int p_off = (K::HPX + (by * 4 + iy) * kTX + bx * 4 + ix) * kKC + sl;
  int q_off[NBR];   // per block row of the window; block column qj adds 4 pixels
#pragma unroll
  for (int qi = 0; qi < NBR; ++qi) q_off[qi] = ((by * 4 + 4 * qi + iy) * K::HC + bx * 4 + ix) * kKC + sl;

  f32x4 acc[K::NB];
#pragma unroll
  for (int i = 0; i < K::NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  if (nch > 1) issue(1);
  for (int ci = 0; ci < nch; ++ci) {
    // chunk ci has landed (mine: at most the newest one may still be in flight) and, past the barrier, everybody's
    if (ci + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K::NIMIN) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ci + 2 < nch) issue(ci + 2);   // stage (ci + 2) % 3 was last read in step ci - 1: free since the barrier
    if (g.ablate & 2) continue;
    const float* st = lds + (ci % 3) * K::STAGE;
    const f32x2 b = *reinterpret_cast<const f32x2*>(st + p_off);
#pragma unroll
    for (int qi = 0; qi < NBR; ++qi) {
      f32x2 a[NBR];
#pragma unroll
      for (int qj = 0; qj < NBR; ++qj) a[qj] = *reinterpret_cast<const f32x2*>(st + q_off[qi] + qj * 4 * kKC);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int qj = 0; qj < NBR; ++qj)
          acc[qi * NBR + qj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[qj][s], b[s], acc[qi * NBR + qj], 0, 0, 0);
    }
  }
  __syncthreads();   // every wave is done with the stages: they become the exchange buffer of the two channel halves
  if (g.ablate & 4) return;

  // ---- the two channel halves of the workgroup meet in LDS (two rounds: the stages hold half of the accumulators)
  constexpr int HALF = (K::NB + 1) / 2;
  static_assert(6 * HALF * 256 * 4 <= (int)K::LDS, "exchange buffer fits the stages");
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int nb0 = round * HALF, nb1 = round ? K::NB : HALF;
    if (kh == 1) {
#pragma unroll
      for (int nb = 0; nb < K::NB; ++nb)
        if (nb >= nb0 && nb < nb1) *reinterpret_cast<f32x4*>(lds + ((pb * HALF + nb - nb0) * 64 + lane) * 4) = acc[nb];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int nb = 0; nb < K::NB; ++nb)
        if (nb >= nb0 && nb < nb1) acc[nb] += *reinterpret_cast<const f32x4*>(lds + ((pb * HALF + nb - nb0) * 64 + lane) * 4);
    }
    __syncthreads();
  }

  NHWC_TRACE(2, wall_clock64());
  if (g.ablate & 16) return;
  // ---- the channel slices of the tile meet in the workspace: slab [(n, tile)][slice][block][nb][lane] of 4 floats
  const long slab_floats = 6l * K::NB * 256;
  float* slab0 = g.slabs + ((long)n * ntiles + tile) * g.ksplit * slab_floats;
  bool reducer = true;
  if (g.ksplit > 1) {
    if (kh == 0) {
      float* mine = slab0 + ks * slab_floats + (long)pb * K::NB * 256;
#pragma unroll
      for (int nb = 0; nb < K::NB; ++nb) {
        // write-through (sc1) stores: 25 KB per wave published without a release fence, i.e. without writing back the
        // whole L2 of this XCD (MI355X_MICROARCH.md "publish-large": 3.0 vs 8.2 us)
        float* dst = mine + (nb * 64 + lane) * 4;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[nb]) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      ticket_s = __hip_atomic_fetch_add(&g.tickets[n * ntiles + tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    reducer = ticket_s == g.ksplit - 1;
    if (!reducer || (g.ablate & 8)) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      g.tickets[n * ntiles + tile] = 0;   // ready for the next launch on this stream
    }
    __syncthreads();
  }

  // ---- window entries of this tile, in the caller's layout.  All 12 waves share the work: wave (pb, kh) takes the
  // window blocks nb of its pixel block with nb % 2 == kh (ksplit == 1: the kh == 0 waves write their own registers).
  const float inv = 1.f / (float)g.C;
  const int jy = li >> 2, jx = li & 3;                       // D[i][j]: j = lane % 16 is the frame-t pixel ...
  const int y = ty0 + by * 4 + jy, x = tx0 + bx * 4 + jx;    // ... i = 4 * (lane / 16) + reg the frame-(t+tau) pixel
  const int py = g.origin + y, pxx = g.origin + x;
  const bool p_in = y < g.oh && x < g.ow;
  const bool p_img = py >= 0 && py < g.H && pxx >= 0 && pxx < g.W;
  // The window entries are first assembled in LDS, [tile pixel][(dy + R) * D + dx + R] (zero where p or q lies in the
  // padding), then streamed out by the whole workgroup in the order of the caller's layout: one contiguous run of D*D
  // floats per pixel for position-major rows, 12-pixel row segments per displacement plane for NCHW.  (Storing straight
  // from the accumulator layout is 4 bytes per lane to 16 different lines per instruction: 21 us per tile.)
  const int D = 2 * g.R + 1, DD = D * D;
  float* tilebuf = lds;
  static_assert(kTY * kTX * (2 * K::R + 1) * (2 * K::R + 1) * 4 <= (int)K::LDS, "output tile fits the stages");
  const int pxi = (by * 4 + jy) * kTX + bx * 4 + jx;
  auto emit = [&](int nb, const f32x4& v) {
    if (g.ablate & 32) return;
    const int qi = nb / NBR, qj = nb % NBR;
    const int dy = 4 * qi + lg - jy - K::R;                  // displacement inside the sub-window (halo origin = tile - K::R)
    if (dy < -g.R || dy > g.R) return;
    const int qyy = py + dy + qy;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dx = 4 * qj + r - jx - K::R;
      if (dx < -g.R || dx > g.R) continue;
      const int qxx = pxx + dx + qx;
      const bool in_image = p_img && qyy >= 0 && qyy < g.H && qxx >= 0 && qxx < g.W;
      tilebuf[pxi * DD + (dy + g.R) * D + dx + g.R] = in_image ? v[r] * inv : 0.f;
    }
  };
  if (g.ksplit > 1) {
    // slice-major: all of this wave's window blocks of one slice are requested together (independent loads), the
    // slices are added in order -- four round trips instead of one per block and slice
    constexpr int MINE = (K::NB + 1) / 2;
    f32x4 v[MINE];
    const float* src = slab0 + (long)pb * K::NB * 256 + lane * 4;
#pragma unroll
    for (int k = 0; k < MINE; ++k) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 1 < g.ksplit; s += 2) {   // two slices in flight, added in slice order
      f32x4 t0[MINE], t1[MINE];
#pragma unroll
      for (int k = 0; k < MINE; ++k) {   // unconditional (a guarded load makes the compiler wait out every single one)
        t0[k] = *reinterpret_cast<const f32x4*>(src + s * slab_floats + min(2 * k + kh, K::NB - 1) * 256);
        t1[k] = *reinterpret_cast<const f32x4*>(src + (s + 1) * slab_floats + min(2 * k + kh, K::NB - 1) * 256);
      }
#pragma unroll
      for (int k = 0; k < MINE; ++k) v[k] = (v[k] + t0[k]) + t1[k];
    }
    for (; s < g.ksplit; ++s) {
#pragma unroll
      for (int k = 0; k < MINE; ++k)
        v[k] += *reinterpret_cast<const f32x4*>(src + s * slab_floats + min(2 * k + kh, K::NB - 1) * 256);
    }
#pragma unroll
    for (int k = 0; k < MINE; ++k)
      if (2 * k + kh < K::NB) emit(2 * k + kh, v[k]);
  } else if (kh == 0) {
#pragma unroll
    for (int nb = 0; nb < K::NB; ++nb) emit(nb, acc[nb]);
  }
  __syncthreads();
  (void)p_in;
  float* ob = g.out + (long)n_img * g.out_sb;
  const int shift_y = qy + g.Rfull - g.R, shift_x = qx + g.Rfull - g.R;
  // (the window size is a compile-time constant on the usual path: the index arithmetic below divides by it per element)
  auto write_out = [&](auto dconst) {
    constexpr int DC = decltype(dconst)::value;
    const int Dv = DC > 0 ? DC : D, DDv = Dv * Dv, total = kTY * kTX * DDv;
    if (g.out_sc == 1) {
      for (int idx = tid; idx < total; idx += kNThreads) {
        const int p = idx / DDv, d = idx - p * DDv;
        const int yy = ty0 + p / kTX, xx = tx0 + p % kTX;
        if (yy >= g.oh || xx >= g.ow) continue;
        const int tj = d / Dv, ti = d - tj * Dv;
        ob[((long)yy * g.ow + xx) * g.out_sp + (long)((tj + shift_y) * g.Dfull + ti + shift_x)] = tilebuf[idx];
      }
    } else {
      for (int idx = tid; idx < total; idx += kNThreads) {
        const int d = idx / (kTY * kTX), p = idx - d * (kTY * kTX);
        const int yy = ty0 + p / kTX, xx = tx0 + p % kTX;
        if (yy >= g.oh || xx >= g.ow) continue;
        const int tj = d / Dv, ti = d - tj * Dv;
        ob[(long)((tj + shift_y) * g.Dfull + ti + shift_x) * g.out_sc + ((long)yy * g.ow + xx) * g.out_sp] = tilebuf[p * DDv + d];
      }
    }
  };
  if (g.R == K::R) write_out(std::integral_constant<int, 2 * K::R + 1>{});
  else write_out(std::integral_constant<int, 0>{});
  NHWC_TRACE(3, wall_clock64());
}

template <int NBR>
int launch_nhwc(NGeom g, hipStream_t stream) {
  using K = NCfg<NBR>;
  static DttDeviceOnce once;
  bool& done = once.here();
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_nhwc_kernel<NBR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS);
    DTT_REQUIRE(e == hipSuccess, "correlation (channels-last): cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    done = true;
  }
  hipLaunchKernelGGL((corr_nhwc_kernel<NBR>), dim3(g.tiles_x * g.tiles_y * g.ksplit * g.batch * g.nquad), dim3(kNThreads), K::LDS, stream, g);
  DTT_CHECK_LAUNCH("corr_nhwc_kernel");
  return 1;
}

struct NPlan { int nbr, ksplit, c_per_split, tiles_x, tiles_y; size_t slab_bytes, ticket_bytes; };

int plan_nhwc(int batch, int C, int oh, int ow, int R, NPlan* p) {
  if (R < 1 || C % kKC != 0) return 0;
  const int r8 = R <= 4 ? 4 : 8;
  p->nbr = r8 == 4 ? 3 : 5;
  p->tiles_x = (ow + kTX - 1) / kTX;
  p->tiles_y = (oh + kTY - 1) / kTY;
  const int tiles = p->tiles_x * p->tiles_y * batch;
  int ncu = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      ncu = prop.multiProcessorCount;
  }
  // one workgroup (12 waves, ~144 KB of LDS) per CU: as many channel slices as fill the chip once
  int ks = ncu / (tiles > 0 ? tiles : 1);
  if (ks < 1) ks = 1;
  const int max_ks = C / (2 * kKC);                    // at least two chunks per slice
  if (ks > max_ks) ks = max_ks > 0 ? max_ks : 1;
  int cps = (C + ks - 1) / ks;
  cps = ((cps + kKC - 1) / kKC) * kKC;
  p->c_per_split = cps;
  p->ksplit = (C + cps - 1) / cps;
  const int nb = p->nbr * p->nbr;
  p->slab_bytes = p->ksplit > 1 ? (size_t)tiles * p->ksplit * 6 * nb * 256 * sizeof(float) : 0;
  // a fixed 64 KB ticket area for anything up to 16384 tiles: geometries of that size share one zero-filled workspace
  p->ticket_bytes = std::max((size_t)65536, ((size_t)tiles * sizeof(int) + 255) & ~(size_t)255);
  return 1;
}

}  // namespace

#ifdef DTT_WG_TRACE
extern "C" int dtt_nhwc_trace_read(unsigned long long* host, int n) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(dtt_nhwc_trace), sizeof(unsigned long long) * n) == hipSuccess;
}
#endif

extern "C" size_t dtt_correlation_nhwc_workspace_bytes(int batch, int ic, int ih, int iw, int pad_size, int kernel_size,
                                                       int max_displacement, int stride1, int stride2) {
  int oc, oh, ow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow)) return 0;
  if (kernel_size != 1 || stride1 != stride2) return 0;
  const int R = max_displacement / stride2;
  NPlan p;
  if (!plan_nhwc(batch * (R > 8 ? 4 : 1), ic, oh, ow, R > 8 ? 8 : R, &p)) return 0;   // 8 < R <= 16: four sub-windows = four virtual images
  return p.slab_bytes + p.ticket_bytes;
}

// WORKSPACE CONTRACT: zero-fill the workspace (hipMemset) once after allocating it; every call leaves its ticket area
// zeroed again, so the same buffer serves any number of calls on ONE stream at a time (calls that may overlap on different
// streams need a workspace each).  There is no per-call memset: it was a 5 us launch in front of a 23-115 us kernel.
// input1 / input2: (batch, ih, iw, ic) channels-last.  Output addressing as dtt_correlation_forward_strided.  Supports
// kernel_size 1, stride1 == stride2 = s, (max_displacement - pad_size) % s == 0, ic % 16 == 0 and max_displacement / s <= 8, or
// 16 with a multiple-of-4 ... (four sub-windows); everything else: transpose and call dtt_correlation_forward_strided.
static int corr_nhwc_ticket_forward(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                    long out_ch_stride, long out_px_stride, const float* input1, int ic, int ih,
                                    int iw, const float* input2, void* workspace, size_t workspace_bytes,
                                    int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                    hipStream_t stream);

extern "C" int dtt_correlation_forward_nhwc_limited(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                                    long out_ch_stride, long out_px_stride, const float* input1, int ic, int ih,
                                                    int iw, const float* input2, void* workspace, size_t workspace_bytes,
                                                    int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                                    int max_workgroups, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(output && input1 && input2, "correlation (channels-last): null pointer");
  int eoc, eoh, eow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &eoc, &eoh, &eow))
    return 0;
  DTT_REQUIRE(ob > 0 && oc == eoc && oh == eoh && ow == eow, "correlation (channels-last): output is (%d,%d,%d,%d), expected (B,%d,%d,%d)",
              ob, oc, oh, ow, eoc, eoh, eow);
  DTT_REQUIRE((((size_t)input1 | (size_t)input2) & 15) == 0, "correlation (channels-last): inputs must be 16-byte aligned");
  // DTT_CORR_NHWC_IMPL=ticket: round 2's channel-split kernel with the in-launch slab reduction (developer A/B)
  static const bool use_ticket = getenv("DTT_CORR_NHWC_IMPL") && !strcmp(getenv("DTT_CORR_NHWC_IMPL"), "ticket");
  if (!use_ticket) {
    DTT_REQUIRE(dtt_corr_wsplit_supported(ic, kernel_size, max_displacement, pad_size, stride1, stride2),
                "correlation (channels-last): needs kernel_size 1, stride1 == stride2 = s, displacement and padding multiples of s, "
                "window radius <= 16 and channels %% 16 == 0");
    return dtt_corr_wsplit_forward(output, ob, oc, oh, ow, out_batch_stride, out_ch_stride, out_px_stride, input1, ic, ih, iw,
                                   input2, pad_size, max_displacement, stride1, max_workgroups, stream);
  }
  return corr_nhwc_ticket_forward(output, ob, oc, oh, ow, out_batch_stride, out_ch_stride, out_px_stride, input1, ic, ih, iw,
                                  input2, workspace, workspace_bytes, pad_size, kernel_size, max_displacement, stride1, stride2,
                                  stream);
}

extern "C" int dtt_correlation_forward_nhwc(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                            long out_ch_stride, long out_px_stride, const float* input1, int ic, int ih,
                                            int iw, const float* input2, void* workspace, size_t workspace_bytes,
                                            int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                            void* stream_) {
  return dtt_correlation_forward_nhwc_limited(output, ob, oc, oh, ow, out_batch_stride, out_ch_stride, out_px_stride, input1, ic,
                                              ih, iw, input2, workspace, workspace_bytes, pad_size, kernel_size, max_displacement,
                                              stride1, stride2, 0, stream_);
}

static int corr_nhwc_ticket_forward(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                    long out_ch_stride, long out_px_stride, const float* input1, int ic, int ih,
                                    int iw, const float* input2, void* workspace, size_t workspace_bytes,
                                    int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                    hipStream_t stream) {
  DTT_REQUIRE(kernel_size == 1 && stride1 == stride2, "correlation (channels-last): kernel_size 1 and stride1 == stride2 only");
  const int s = stride1, Rfull = max_displacement / s;
  DTT_REQUIRE((max_displacement - pad_size) % s == 0 && max_displacement % s == 0,
              "correlation (channels-last): displacement and padding must be multiples of the stride");
  DTT_REQUIRE(ic % kKC == 0, "correlation (channels-last): channels (%d) must be a multiple of %d", ic, kKC);
  DTT_REQUIRE(Rfull >= 1 && (Rfull <= 8 || (Rfull <= 16 && Rfull % 4 == 0)), "correlation (channels-last): window radius %d not supported", Rfull);
  DTT_REQUIRE((((size_t)input1 | (size_t)input2) & 15) == 0, "correlation (channels-last): inputs must be 16-byte aligned");
  NGeom g;
  g.f1 = input1; g.f2 = input2;
  g.C = ic;
  g.H = (ih + s - 1) / s; g.W = (iw + s - 1) / s;            // lattice pixels 0, s, 2s, ...
  g.sx = (long)s * ic; g.sy = (long)s * iw * ic; g.sb = (long)ih * iw * ic;
  g.oh = oh; g.ow = ow; g.origin = (max_displacement - pad_size) / s;
  g.Rfull = Rfull; g.Dfull = 2 * Rfull + 1;
  g.batch = ob;
  g.ablate = getenv("DTT_CORR_NHWC_ABLATE") ? atoi(getenv("DTT_CORR_NHWC_ABLATE")) : 0;
  g.trace_slot = -1;
#ifdef DTT_WG_TRACE
  static int trace_launches = 0;
  if (ic == 2048) g.trace_slot = trace_launches++ % 16;
#endif
  g.out = output; g.out_sb = out_batch_stride; g.out_sc = out_ch_stride; g.out_sp = out_px_stride;
  const int R = Rfull > 8 ? 8 : Rfull;
  const int nquad = Rfull > 8 ? 4 : 1;
  NPlan p;
  DTT_REQUIRE(plan_nhwc(ob * nquad, ic, oh, ow, R, &p), "correlation (channels-last): no plan");
  DTT_REQUIRE(workspace && workspace_bytes >= p.slab_bytes + p.ticket_bytes, "correlation (channels-last): workspace too small (%zu < %zu)",
              workspace_bytes, p.slab_bytes + p.ticket_bytes);
  g.R = R; g.tiles_x = p.tiles_x; g.tiles_y = p.tiles_y; g.ksplit = p.ksplit; g.c_per_split = p.c_per_split;
  g.tickets = static_cast<int*>(workspace);
  g.slabs = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.ticket_bytes);
  // (the tickets are zero on entry -- workspace contract -- and the reducer of every tile zeroes its own again)
  dtt_prof_begin("corr_fwd_op", stream);
  dtt_prof_begin("corr_nhwc", stream);
  // 8 < R <= 16: four (2*8+1)^2 sub-windows centred at (+-(R-8), +-(R-8)), overlapping rows / columns being the same
  // arithmetic written twice -- run as four "virtual images" of ONE launch (the channel split is planned for 4 x batch
  // images: at batch 1 two slices of 64 chunks per sub-window instead of eight slices of 16 in each of four launches)
  g.nquad = nquad;
  const int c = Rfull - 8;
  for (int q = 0; q < 4; ++q) {
    g.qyv[q] = nquad == 1 ? 0 : ((q >> 1) ? c : -c);
    g.qxv[q] = nquad == 1 ? 0 : ((q & 1) ? c : -c);
  }
  const int ok = p.nbr == 3 ? launch_nhwc<3>(g, stream) : launch_nhwc<5>(g, stream);
  dtt_prof_end("corr_nhwc", stream);
  dtt_prof_end("corr_fwd_op", stream);
  return ok;
}

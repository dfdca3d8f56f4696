// FlowNet-style cross-frame correlation for gfx950.
//
// Replaces channels_first + Correlation_forward / Correlation_backward_input{1,2} (reference
// correlation/src/correlation_cuda_kernel.cu:10-106, 108-290).  The reference first repacks both feature
// maps into zero-padded NHWC scratch (after zero-filling it), then runs one 32-thread block per output
// pixel that walks all D^2 displacements serially, re-reading frame t from L2 D^2 times.
//
// Here, for kernel_size 1 and stride1 == stride2 (every D&T call site, rfcn.py:58-60), the op is a banded
// matrix product:   out[p, q] = sum_c f1[c, p] * f2[c, q]   for |q - p| <= R on the stride lattice,
// computed with exact-f32 MFMA (v_mfma_f32_16x16x4_f32 is bitwise an fmaf chain, so no precision is
// traded).  One workgroup = one 8x8 tile of output pixels x one channel slice:
//   * NCHW is read directly (no repack, no scratch fills);
//   * per 8-channel chunk the 8x8 frame-t tile and the (8+2R)^2 frame-(t+tau) halo are staged in LDS
//     as [c][row][col] with row / plane strides chosen so both MFMA operand reads are conflict free;
//   * each of the 4 waves owns a 4x4 block of frame-t pixels (MFMA rows) and multiplies it against the
//     (4+2R)^2 halo it needs, cut into 4x4 pixel blocks (MFMA columns): 72 % of the MFMA work lands
//     inside the displacement window at R = 8;
//   * accumulators stay in registers for the whole channel loop; partial sums of the channel slices go
//     to a workspace in fragment order (coalesced) and a small reduce kernel sums the slices in a fixed
//     order, divides by C and scatters into NCHW -- deterministic, no atomics.
// Kernels, fastest first (all produce the same partials layout):
//   corr_fwd_glds     stride 1, 16-byte aligned halo columns: LDS-DMA staging (global_load_lds_dwordx4), double
//                     buffered, zero padding applied by the reduce kernel -- the D&T conv4 / conv5 path;
//   corr_fwd_mfma_v4  stride 1 otherwise: 16-byte register staging, padding by masks at the LDS write;
//   corr_fwd_mfma     any stride (conv3): scalar register staging, software-pipelined;
//   corr_fwd_generic  kernel_size > 1 or stride1 != stride2: wave per output pixel.
// Backward: corr_bwd_mfma (band in registers, other frame streamed through LDS), corr_bwd_simple fallback.
// DTT_CORR_ABLATE / DTT_CORR_STAMP / DTT_CORR_CPHASE are developer builds for tools/corr_tune.py (timing
// experiments and the per-workgroup timeline); they never ship in libdtt_hip.so.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef DTT_CORR_ABLATE
#define DTT_CORR_ABLATE 0     // developer ablations for tools/corr_tune.py: 1 = no global loads, 2 = also no LDS
#endif                        // writes, 3 = also no barriers (results are wrong for any value but 0)
#ifndef DTT_CORR_KC
#define DTT_CORR_KC 8
#endif
#ifndef DTT_CORR_MINW
#define DTT_CORR_MINW 3
#endif
#ifndef DTT_CORR_PF
#define DTT_CORR_PF 6
#endif
constexpr int kKc = DTT_CORR_KC;  // channels per LDS chunk (register-staged kernels)
#ifndef DTT_CORR_GKC
#define DTT_CORR_GKC 8
#endif
constexpr int kGKc = DTT_CORR_GKC;  // channels per LDS chunk of the LDS-DMA kernel (4 wins a back-to-back micro-benchmark by 10 %, 8 wins inside the pipeline by 2 %)
constexpr int kTile = 8;      // output tile edge (lattice pixels)
constexpr int kThreads = 256; // 4 waves, one 4x4 M-block each
constexpr int kPS1 = 68;      // frame-t plane stride (64 px, +4: second k pair lands on the other banks)

struct FastGeom {
  int C, H, W;          // input
  int oc, oh, ow;       // output
  int s;                // lattice step (= stride1 = stride2)
  int origin;           // unpadded coordinate of lattice point 0 (= max_displacement - pad_size)
  int R;                // displacement radius in lattice units
  int D;                // 2R + 1
  int tiles_x, tiles_y; // output tiles
  int ksplit, c_per_split;
  // window decomposition (LDS-DMA kernel + reduce only): this launch computes the (2R+1)^2 sub-window centred at
  // displacement (qy, qx) of a larger (2*Rfull+1)^2 window; plain launches have qy = qx = 0, Rfull = R, Dfull = D
  int qy, qx, Rfull, Dfull;
};

template <int NBR>
struct Cfg {
  static constexpr int HR = 4 + 4 * NBR;                       // halo rows = cols held in LDS
  static constexpr int HRS = (HR % 32 == 8 || HR % 32 == 24) ? HR : HR + 8;  // row stride: 4 rows on 4 bank groups
  static constexpr int PS2 = ((HR * HRS + 7) / 8) * 8 + 4;     // plane stride = 4 mod 8
  static constexpr int NB = NBR * NBR;                         // N-blocks per wave
  static constexpr int ELEMS = (HR * HR + kThreads - 1) / kThreads;
  static constexpr size_t LDS = (size_t)kKc * (PS2 + kPS1) * sizeof(float) + 16;  // + spare slot for idle lanes
};

// grid: tiles * ksplit * batch (1-D).  ws layout: [ksplit][batch][tile][wave][nb][reg][lane].
// Software pipeline per channel chunk: the global loads of chunk i+1 are issued (unconditionally, to
// clamped addresses) before the MFMA phase of chunk i and land in registers while the matrix pipe works;
// they are written to LDS after the phase's barrier.
template <int NBR, bool PIPE, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void corr_fwd_mfma(const float* __restrict__ in1,
                                                                const float* __restrict__ in2,
                                                                float* __restrict__ ws, FastGeom g) {
  using K = Cfg<NBR>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* l2 = lds;                  // [kKc][PS2]
  float* l1 = lds + kKc * K::PS2;   // [kKc][kPS1]

  // 1-D grid of tiles x channel slices x batch.  Work items are ordered (batch, slice, tile) and each XCD gets a
  // contiguous range of them, so the ~9x halo re-reads of one channel slice (a few MB) hit that XCD's own L2.
  const int ntiles = g.tiles_x * g.tiles_y;
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntiles;
  const int ks = (item / ntiles) % g.ksplit, n = item / (ntiles * g.ksplit);
  const int nbatch = gridDim.x / (ntiles * g.ksplit);
  const int ty0 = (tile / g.tiles_x) * kTile, tx0 = (tile % g.tiles_x) * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int my = wave >> 1, mx = wave & 1;
  const int plane = g.H * g.W;
  const int c_begin = ks * g.c_per_split;
  const int c_end = min(g.C, c_begin + g.c_per_split);

  // ---- per-thread staging descriptors (identical for every channel)
  int goff2[K::ELEMS], loff2[K::ELEMS];
  bool ok2[K::ELEMS];
#pragma unroll
  for (int e = 0; e < K::ELEMS; ++e) {
    const int idx = tid + e * kThreads;
    const int hr = idx / K::HR, hc = idx - hr * K::HR;
    const int gy = g.origin + (ty0 - g.R + hr) * g.s, gx = g.origin + (tx0 - g.R + hc) * g.s;
    const bool in_lds = idx < K::HR * K::HR;
    ok2[e] = in_lds && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
    goff2[e] = ok2[e] ? gy * g.W + gx : 0;
    loff2[e] = in_lds ? hr * K::HRS + hc : K::PS2 - 1;  // surplus threads dump into the plane's unused pad word
  }
  int goff1;
  bool ok1;
  const int ch1 = tid >> 6;  // 4 channels per pass
  {
    const int p = tid & 63;
    const int py = p >> 3, px = p & 7;
    const int gy = g.origin + (ty0 + py) * g.s, gx = g.origin + (tx0 + px) * g.s;
    ok1 = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && (ty0 + py) < g.oh && (tx0 + px) < g.ow;
    goff1 = ok1 ? gy * g.W + gx : 0;
  }
  const int loff1 = tid & 63;
  unsigned m2[K::ELEMS];
#pragma unroll
  for (int e = 0; e < K::ELEMS; ++e) m2[e] = ok2[e] ? 0xFFFFFFFFu : 0u;
  const unsigned m1 = ok1 ? 0xFFFFFFFFu : 0u;

  // ---- MFMA operand addresses
  const int kq_lane = lane >> 4;               // k index inside a quad
  const int ij = lane & 15;
  const int a_off = kq_lane * kPS1 + (my * 4 + (ij >> 2)) * kTile + mx * 4 + (ij & 3);
  const int b_off = kq_lane * K::PS2 + (my * 4 + (ij >> 2)) * K::HRS + mx * 4 + (ij & 3);

  f32x4 acc[K::NB];
#pragma unroll
  for (int i = 0; i < K::NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* p1 = in1 + (long)n * g.C * plane;
  const float* p2 = in2 + (long)n * g.C * plane;

  auto mfma_phase = [&]() {
#pragma unroll
    for (int kq = 0; kq < kKc / 4; ++kq) {
      const float a = l1[kq * 4 * kPS1 + a_off];
      const float* bp = l2 + kq * 4 * K::PS2 + b_off;
#pragma unroll
      for (int nby = 0; nby < NBR; ++nby)
#pragma unroll
        for (int nbx = 0; nbx < NBR; ++nbx) {
          const float b = bp[nby * 4 * K::HRS + nbx * 4];
          acc[nby * NBR + nbx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nby * NBR + nbx], 0, 0, 0);
        }
    }
  };

  if constexpr (PIPE) {
    float st2[kKc][K::ELEMS];
    float st1[kKc / 4];
    auto issue_loads = [&](int c0) {
#pragma unroll
      for (int cc = 0; cc < kKc; ++cc) {
        const int c = min(c0 + cc, g.C - 1);  // clamped: out-of-slice channels are zeroed at write time
        const float* src = p2 + (long)c * plane;
#pragma unroll
        for (int e = 0; e < K::ELEMS; ++e) st2[cc][e] = src[goff2[e]];
      }
#pragma unroll
      for (int q = 0; q < kKc / 4; ++q) {
        const int c = min(c0 + q * 4 + ch1, g.C - 1);
        st1[q] = p1[(long)c * plane + goff1];
      }
    };
    auto write_lds = [&](int c0) {
      // branch-free: zero padding / channel tail by AND-mask, every thread stores every element
#pragma unroll
      for (int cc = 0; cc < kKc; ++cc) {
        const unsigned cmask = (c0 + cc) < c_end ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int e = 0; e < K::ELEMS; ++e)
          l2[cc * K::PS2 + loff2[e]] = __uint_as_float(__float_as_uint(st2[cc][e]) & (cmask & m2[e]));
      }
#pragma unroll
      for (int q = 0; q < kKc / 4; ++q) {
        const int cc = q * 4 + ch1;
        const unsigned cmask = (c0 + cc) < c_end ? 0xFFFFFFFFu : 0u;
        l1[cc * kPS1 + loff1] = __uint_as_float(__float_as_uint(st1[q]) & (cmask & m1));
      }
    };
#if DTT_CORR_ABLATE == 0
    issue_loads(c_begin);
    write_lds(c_begin);
    __syncthreads();
    for (int c0 = c_begin; c0 < c_end; c0 += kKc) {
      const bool more = c0 + kKc < c_end;
      if (more) issue_loads(c0 + kKc);
      mfma_phase();
      __syncthreads();
      if (more) {
        write_lds(c0 + kKc);
        __syncthreads();
      }
    }
#else
#pragma unroll
    for (int cc = 0; cc < kKc; ++cc)
#pragma unroll
      for (int e = 0; e < K::ELEMS; ++e) st2[cc][e] = (float)(tid + cc + e);
#pragma unroll
    for (int q = 0; q < kKc / 4; ++q) st1[q] = (float)(tid + q);
    write_lds(c_begin);
    __syncthreads();
    for (int c0 = c_begin; c0 < c_end; c0 += kKc) {
      const bool more = c0 + kKc < c_end;
      mfma_phase();
      if (DTT_CORR_ABLATE < 3) __syncthreads();
      if (more) {
        if (DTT_CORR_ABLATE < 2) write_lds(c0 + kKc);
        if (DTT_CORR_ABLATE < 3) __syncthreads();
      }
    }
    (void)issue_loads;
#endif
  } else {
    // register-lean variant (large displacement windows): stage 4 channels at a time, no prefetch
    for (int c0 = c_begin; c0 < c_end; c0 += kKc) {
#pragma unroll 1
      for (int cq = 0; cq < kKc; cq += 4) {
        float st[4][K::ELEMS];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const float* src = p2 + (long)min(c0 + cq + cc, g.C - 1) * plane;
#pragma unroll
          for (int e = 0; e < K::ELEMS; ++e) st[cc][e] = src[goff2[e]];
        }
        const float v1 = p1[(long)min(c0 + cq + ch1, g.C - 1) * plane + goff1];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const unsigned cmask = (c0 + cq + cc) < c_end ? 0xFFFFFFFFu : 0u;
#pragma unroll
          for (int e = 0; e < K::ELEMS; ++e)
            l2[(cq + cc) * K::PS2 + loff2[e]] = __uint_as_float(__float_as_uint(st[cc][e]) & (cmask & m2[e]));
        }
        l1[(cq + ch1) * kPS1 + loff1] =
            __uint_as_float(__float_as_uint(v1) & (((c0 + cq + ch1) < c_end ? 0xFFFFFFFFu : 0u) & m1));
      }
      __syncthreads();
      mfma_phase();
      __syncthreads();
    }
  }

  // ---- partial sums to the workspace, fragment order (each store instruction = 256 contiguous bytes)
  float* w = ws + ((((long)ks * nbatch + n) * ntiles + tile) * 4 + wave) * (long)(K::NB * 256);
#pragma unroll
  for (int nb = 0; nb < K::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) w[(nb * 4 + r) * 64 + lane] = acc[nb][r];
}

// Stride-1 variant of the kernel above with 16-byte staging: every thread moves float4 groups
// (global_load_dwordx4 at 4-byte alignment -- gfx950 runs in unaligned-access mode -- and ds_write_b128), 10 loads
// + 10 LDS stores per thread per chunk instead of 52 + 52.  Requires C % 16 == 0 (no channel tail) and W >= 4.
// Groups that straddle the left / right image edge are loaded from the clamped in-row position and rotated
// into place (workgroup-uniform slow path); out-of-image components are zeroed by mask.
template <int NBR, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void corr_fwd_mfma_v4(const float* __restrict__ in1,
                                                                   const float* __restrict__ in2,
                                                                   float* __restrict__ ws, FastGeom g) {
  using K = Cfg<NBR>;
#ifdef DTT_CORR_STAMP
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int G4 = K::HR / 4;             // float4 groups per halo row
  constexpr int PL4 = K::HR * G4;           // groups per channel plane
  constexpr int N4 = (kKc * PL4 + kThreads - 1) / kThreads;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* l2 = lds;                  // [kKc][PS2]
  float* l1 = lds + kKc * K::PS2;   // [kKc][kPS1]

  const int ntiles = g.tiles_x * g.tiles_y;
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntiles;
  const int ks = (item / ntiles) % g.ksplit, n = item / (ntiles * g.ksplit);
  const int nbatch = gridDim.x / (ntiles * g.ksplit);
  const int ty0 = (tile / g.tiles_x) * kTile, tx0 = (tile % g.tiles_x) * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int my = wave >> 1, mx = wave & 1;
  const int plane = g.H * g.W;
  const int c_begin = ks * g.c_per_split;
  const int c_end = min(g.C, c_begin + g.c_per_split);

  // descriptor of one float4 group: global offset (channel-in-chunk included), LDS offset, 4 validity bits,
  // rotation delta = wanted_start - loaded_start
  auto describe = [&](int cc, int gy, int start, bool live, int& goff, unsigned& bits, int& delta) {
    const bool row_ok = live && gy >= 0 && gy < g.H;
    bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) bits |= (row_ok && start + j >= 0 && start + j < g.W) ? (1u << j) : 0u;
    const int ld = min(max(start, 0), g.W - 4);
    delta = bits ? start - ld : 0;
    goff = bits ? cc * plane + gy * g.W + ld : 0;
  };
  int goff2[N4], loff2[N4], dl2[N4];
  unsigned vb2[N4];
  bool fix = false;
#pragma unroll
  for (int i = 0; i < N4; ++i) {
    const int f = tid + i * kThreads;
    const bool live = f < kKc * PL4;
    const int cc = live ? f / PL4 : 0, r = live ? f - cc * PL4 : 0;
    const int hr = r / G4, q = r - hr * G4;
    describe(cc, g.origin + ty0 - g.R + hr, g.origin + tx0 - g.R + 4 * q, live, goff2[i], vb2[i], dl2[i]);
    loff2[i] = live ? cc * K::PS2 + hr * K::HRS + 4 * q : kKc * K::PS2 + kKc * kPS1;  // dead lanes -> spare slot
    fix |= dl2[i] != 0;
  }
  int goff1, loff1, dl1;
  unsigned vb1;
  {
    // 16 float4 groups per channel: with kKc = 8 only threads 0..127 own a group of the frame-t tile.  (The others used
    // to stage channels kKc..15 -- past the chunk, and past the end of the tensor on its last chunk: tools/fuzz_ops.py.)
    const int cc = tid >> 4, r = tid & 15, py = r >> 1, q = r & 1;
    const bool in_chunk = cc < kKc;
    const bool live = in_chunk && (ty0 + py) < g.oh;
    describe(in_chunk ? cc : 0, g.origin + ty0 + py, g.origin + tx0 + 4 * q, live, goff1, vb1, dl1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tx0 + 4 * q + j >= g.ow) vb1 &= ~(1u << j);
    loff1 = in_chunk ? cc * kPS1 + py * kTile + 4 * q : kKc * kPS1;   // idle threads: spare slot behind the tile
    fix |= dl1 != 0;
  }
  const bool wg_fix = __syncthreads_or(fix);

  const int kq_lane = lane >> 4;
  const int ij = lane & 15;
  const int a_off = kq_lane * kPS1 + (my * 4 + (ij >> 2)) * kTile + mx * 4 + (ij & 3);
  const int b_off = kq_lane * K::PS2 + (my * 4 + (ij >> 2)) * K::HRS + mx * 4 + (ij & 3);

  f32x4 acc[K::NB];
#pragma unroll
  for (int i = 0; i < K::NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* p1 = in1 + (long)n * g.C * plane;
  const float* p2 = in2 + (long)n * g.C * plane;

  f32x4 st2[N4], st1;

  auto load16 = [](const float* p) -> f32x4 {  // 16-byte load at 4-byte alignment (global_load_dwordx4)
    f32x4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
  };
  auto issue_loads = [&](int c0) {
    const float* b2 = p2 + (long)c0 * plane;
    const float* b1 = p1 + (long)c0 * plane;
#pragma unroll
    for (int i = 0; i < N4; ++i) st2[i] = load16(b2 + goff2[i]);
    st1 = load16(b1 + goff1);
  };
  auto place = [&](const f32x4 v, unsigned bits, int delta, bool rotate) -> f32x4 {
    f32x4 o = v;
    if (rotate) {  // wanted component j = loaded component j + delta
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = j + delta;
        float x = v[0];
        x = k == 1 ? v[1] : x;
        x = k == 2 ? v[2] : x;
        x = k == 3 ? v[3] : x;
        o[j] = x;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = __uint_as_float(__float_as_uint(o[j]) & (unsigned)(-(int)((bits >> j) & 1u)));
    return o;
  };
  auto write_lds = [&]() {
    if (wg_fix) {  // one workgroup-uniform branch around the whole store pass
#pragma unroll
      for (int i = 0; i < N4; ++i) *reinterpret_cast<f32x4*>(l2 + loff2[i]) = place(st2[i], vb2[i], dl2[i], true);
      *reinterpret_cast<f32x4*>(l1 + loff1) = place(st1, vb1, dl1, true);
    } else {
#pragma unroll
      for (int i = 0; i < N4; ++i) *reinterpret_cast<f32x4*>(l2 + loff2[i]) = place(st2[i], vb2[i], 0, false);
      *reinterpret_cast<f32x4*>(l1 + loff1) = place(st1, vb1, 0, false);
    }
  };
  // MFMA phase: kKc/4 k-quads x NB column blocks, flattened, with the B operand read PF MFMAs ahead of its use
  // (the compiler otherwise issues ds_read -> s_waitcnt lgkmcnt(0) -> mfma back to back and exposes the LDS
  // latency on every pair).
  auto mfma_phase = [&]() {
    constexpr int TOT = (kKc / 4) * K::NB;
    constexpr int PF = 4;
    auto b_at = [&](int t) -> float {
      const int kq = t / K::NB, nb = t - kq * K::NB;
      return l2[kq * 4 * K::PS2 + b_off + (nb / NBR) * 4 * K::HRS + (nb % NBR) * 4];
    };
    float ring[PF];
#pragma unroll
    for (int t = 0; t < PF; ++t) ring[t] = b_at(t);
    float a = l1[a_off];
#pragma unroll
    for (int t = 0; t < TOT; ++t) {
      const int kq = t / K::NB, nb = t - kq * K::NB;
      const float b = ring[t % PF];
      if (t + PF < TOT) ring[t % PF] = b_at(t + PF);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nb], 0, 0, 0);
      if (nb == K::NB - 1 && kq + 1 < kKc / 4) a = l1[(kq + 1) * 4 * kPS1 + a_off];
    }
  };

  issue_loads(c_begin);
  write_lds();
  __syncthreads();
  for (int c0 = c_begin; c0 < c_end; c0 += kKc) {
    const bool more = c0 + kKc < c_end;
#if DTT_CORR_ABLATE == 4   // all chunks re-read the first chunk's channels: L2-resident working set
    if (more) issue_loads(c_begin);
#elif DTT_CORR_ABLATE == 5 // no global loads at all (LDS rewritten with stale registers)
#else
    if (more) issue_loads(c0 + kKc);
#endif
    mfma_phase();
    __syncthreads();
    if (more) {
#if DTT_CORR_ABLATE != 6   // 6: loads issued but never consumed / no LDS writes
      write_lds();
#endif
      __syncthreads();
    }
  }

  float* w = ws + ((((long)ks * nbatch + n) * ntiles + tile) * 4 + wave) * (long)(K::NB * 256);
#pragma unroll
  for (int nb = 0; nb < K::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) w[(nb * 4 + r) * 64 + lane] = acc[nb][r];
#ifdef DTT_CORR_STAMP  // developer timeline: per-workgroup start / end (100 MHz clock) + HW_ID / XCC_ID after the partials
  if (tid == 0) {
    unsigned long long* st = reinterpret_cast<unsigned long long*>(ws + (long)g.ksplit * nbatch * ntiles * 4 * K::NB * 256) +
                             4 * (long)blockIdx.x;
    st[0] = stamp0;
    st[1] = __builtin_amdgcn_s_memrealtime();
    st[2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    st[3] = (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xF) | ((unsigned long long)item << 8) | ((unsigned long long)wg_fix << 40);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant (stride 1, halo columns 16-byte aligned in the lattice): the staging pass disappears.
// Every lane moves 16-byte pieces global -> LDS with global_load_lds_dwordx4 (no staging VGPRs, no ds_write,
// no address / mask VALU in the loop), two LDS buffers, ONE barrier per chunk: the DMA of chunk i+1 runs
// underneath the MFMA phase of chunk i.  Zero padding moves to the OUTPUT side: every accumulator element
// belongs to exactly one (p, q) pixel pair, so pieces that fall outside the image are simply loaded from a
// clamped in-bounds address (their products are garbage) and corr_fwd_reduce writes 0 for pairs whose p or q
// lies in the padding.  Pieces are addressed flat in the plane, so a piece straddling the right edge reads
// the first pixels of the next row / plane (garbage lanes only); the one place where that would leave the
// tensor (last row of the last plane) is loaded shifted back and rotated in LDS (workgroup-uniform rare path).
// LDS image = piece-linear (DMA destination is wave base + lane * 16):  [c][HR rows][RS] + pad per plane,
// frame-t tile [c][8][8] + pad; strides chosen so that both MFMA operand reads stay bank-conflict free.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) float*)p;
}
template <int OFF>
__device__ __forceinline__ void lds_read_async(float& dst, unsigned addr) {  // completion is counted by hand below
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(float& v) {  // s_waitcnt tied to the value it releases
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

// One MFMA phase over a staged chunk with a hand-counted LDS read-ahead: hipcc sinks every ds_read next to the MFMA
// that consumes it and waits lgkmcnt(0), exposing the LDS latency on each pair.  Here the B operand of MFMA t is
// requested PF-1 (or PF) MFMAs early (asm ds_read the compiler does not track), released by a counted s_waitcnt (LDS
// returns in order: lgkmcnt(n) = "all but the newest n have landed"), and the ring slot of MFMA t-1 is refilled
// after MFMA t has issued.  sched_barrier pins that order.
template <class K, int NBR, int PF, int T>
__device__ __forceinline__ void corr_mfma_steps(f32x4 (&acc)[K::NB], float (&ring)[PF], float (&a)[kGKc / 4],
                                                unsigned baddr) {
  constexpr int TOT = (kGKc / 4) * K::NB;
  constexpr int kq = T / K::NB, nb = T % K::NB;
  constexpr int newer = (PF - 2 < TOT - 1 - T) ? PF - 2 : TOT - 1 - T;
  lds_wait<newer>(ring[T % PF]);
  acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq], ring[T % PF], acc[nb], 0, 0, 0);
  if constexpr (T >= 1 && T - 1 + PF < TOT) {
    constexpr int u = T - 1 + PF, ukq = u / K::NB, unb = u % K::NB;
    lds_read_async<(ukq * 4 * K::PS2 + (unb / NBR) * 4 * K::RS + (unb % NBR) * 4) * 4>(ring[u % PF], baddr);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (T + 1 < TOT) corr_mfma_steps<K, NBR, PF, T + 1>(acc, ring, a, baddr);
}
template <class K, int NBR, int PF, int T>
__device__ __forceinline__ void corr_mfma_prime(float (&ring)[PF], unsigned baddr) {
  constexpr int kq = T / K::NB, nb = T % K::NB;
  lds_read_async<(kq * 4 * K::PS2 + (nb / NBR) * 4 * K::RS + (nb % NBR) * 4) * 4>(ring[T], baddr);
  if constexpr (T + 1 < PF) corr_mfma_prime<K, NBR, PF, T + 1>(ring, baddr);
}

template <int NBR>
struct GCfg {
  static constexpr int HR = 4 + 4 * NBR;                                        // halo rows = cols
  static constexpr int G4 = HR / 4;                                             // real pieces per halo row
  static constexpr int RS = (HR % 32 == 8 || HR % 32 == 24) ? HR : HR + 4;      // row stride (floats)
  static constexpr int R4 = RS / 4;                                             // pieces per row incl. pad
  static constexpr int PS2 = HR * RS + (RS == HR ? 4 : 16);                     // plane stride: 4 mod 8 / 16 mod 32
  static constexpr int PP2 = PS2 / 4;                                           // pieces per halo plane
  static constexpr int PS1 = kTile * kTile + 4, PP1 = PS1 / 4;                  // frame-t plane
  static constexpr int HSLOTS = (kGKc * PP2 + 63) / 64;                          // wave slots of the halo planes (+ pad):
  static constexpr int A0 = HSLOTS * 64 * 4;                                    // the frame-t planes start on a slot boundary
  static constexpr int NP = HSLOTS * 64 + kGKc * PP1;                            // pieces per buffer
  static constexpr int NI = (NP + kThreads - 1) / kThreads;                     // DMA instructions per thread
  static constexpr int BUF = NP * 4;                                            // floats per buffer
  static constexpr int NB = NBR * NBR;
  static constexpr size_t LDS = 2 * (size_t)BUF * sizeof(float);
};

template <int NBR, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void corr_fwd_glds(const float* __restrict__ in1,
                                                                const float* __restrict__ in2,
                                                                float* __restrict__ ws, FastGeom g) {
  using K = GCfg<NBR>;
#ifdef DTT_CORR_STAMP
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ntiles = g.tiles_x * g.tiles_y;
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntiles;
  const int ks = (item / ntiles) % g.ksplit, n = item / (ntiles * g.ksplit);
  const int nbatch = gridDim.x / (ntiles * g.ksplit);
  const int ty0 = (tile / g.tiles_x) * kTile, tx0 = (tile % g.tiles_x) * kTile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my = wave >> 1, mx = wave & 1;
  const int plane = g.H * g.W;
  const int c_begin = ks * g.c_per_split;
  const int c_end = min(g.C, c_begin + g.c_per_split);
  const int nch = (c_end - c_begin) / kGKc;

  // piece p of a buffer -> source: which frame, channel in chunk, in-plane flat offset, overflow past the plane end
  auto describe = [&](int p, bool& is_a, int& cc, int& off, int& over) {
    is_a = p >= K::HSLOTS * 64;
    int gy, start;
    bool real;
    if (!is_a) {
      cc = min(p / K::PP2, kGKc - 1);
      const int r = p - cc * K::PP2, hr = r / K::R4, q = r - hr * K::R4;
      real = p < kGKc * K::PP2 && hr < K::HR && q < K::G4;
      gy = g.origin + ty0 - g.R + hr + g.qy;
      start = g.origin + tx0 - g.R + 4 * q + g.qx;
    } else {
      const int pa = p - K::HSLOTS * 64;
      cc = pa / K::PP1;
      const int r = pa - cc * K::PP1, py = r >> 1, q = r & 1;
      real = r < 2 * kTile;
      gy = g.origin + ty0 + py;
      start = g.origin + tx0 + 4 * q;
    }
    const bool valid = real && gy >= 0 && gy < g.H && start >= 0 && start < g.W;  // holds at least one in-image pixel
    const int gyc = min(max(gy, 0), g.H - 1);
    off = real ? gyc * g.W + start : 0;
    if (!valid) off = min(max(off, 0), plane - 4);
    over = valid ? max(off + 4 - plane, 0) : 0;
  };
  // Source of piece i of this thread = wave-uniform base (frame, image, first channel of the chunk: SGPRs, advanced once
  // per chunk) + a per-lane 32-bit offset that never changes: the DMA instructions take the SGPR-base form and the loop
  // carries no per-lane address arithmetic.
  unsigned soff[K::NI];
#pragma unroll
  for (int i = 0; i < K::NI; ++i) {
    const int p = min(i * kThreads + tid, K::NP - 1);
    bool is_a;
    int cc, off, over;
    describe(p, is_a, cc, off, over);
    soff[i] = (unsigned)(cc * plane + off) * 4u;
  }
  const long chunk_stride = (long)kGKc * plane;
  const float* base2 = in2 + ((long)n * g.C + c_begin) * plane;
  const float* base1 = in1 + ((long)n * g.C + c_begin) * plane;
  // the only pieces that could read past the end of the tensor: last image, last channel, bottom-right straddle
  const bool tail_wg = n == nbatch - 1 && c_end == g.C;

  auto issue = [&](float* buf, bool last_chunk) {
#pragma unroll
    for (int i = 0; i < K::NI; ++i) {
      const int p = i * kThreads + tid;
      unsigned o = soff[i];
      if (tail_wg && last_chunk) {  // workgroup-uniform, once per launch for 1 / (ksplit * batch) of the workgroups
        bool is_a;
        int cc, off, over;
        describe(min(p, K::NP - 1), is_a, cc, off, over);
        if (cc == kGKc - 1) o -= 4u * (unsigned)over;
      }
      const bool slot_is_a = i * 4 + wave >= K::HSLOTS;  // wave-uniform
      const char* sbase = reinterpret_cast<const char*>(slot_is_a ? base1 : base2);
      lds_void_t* dst = (lds_void_t*)(buf + (i * kThreads + wave * 64) * 4);
      if (i * kThreads + kThreads <= K::NP || p < K::NP)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(sbase + o), dst, 16, 0, 0);
    }
    base1 += chunk_stride;
    base2 += chunk_stride;
  };
  auto fix_tail = [&](float* buf) {  // rotate the shifted pieces of the last plane into place
    __syncthreads();
#pragma unroll
    for (int i = 0; i < K::NI; ++i) {
      const int p = i * kThreads + tid;
      bool is_a;
      int cc, off, over;
      describe(min(p, K::NP - 1), is_a, cc, off, over);
      if (p < K::NP && cc == kGKc - 1 && over > 0) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = buf[4 * p + j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j + over < 4) buf[4 * p + j] = v[j + over];
      }
    }
  };

  // a wave whose whole 4x4 pixel block lies beyond the output (right / bottom overhang of the tiling) only helps staging
  const bool wave_live = ty0 + my * 4 < g.oh && tx0 + mx * 4 < g.ow;
  const int kq_lane = lane >> 4;
  const int ij = lane & 15;
  const int a_off = K::A0 + kq_lane * K::PS1 + (my * 4 + (ij >> 2)) * kTile + mx * 4 + (ij & 3);
  const int b_off = kq_lane * K::PS2 + (my * 4 + (ij >> 2)) * K::RS + mx * 4 + (ij & 3);

  f32x4 acc[K::NB];
#pragma unroll
  for (int i = 0; i < K::NB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = lds_byte_addr(lds);
#ifdef DTT_CORR_CPHASE   // compiler-scheduled operand reads (experiment)
  auto mfma_phase = [&](const float* buf) {
    constexpr int TOT = (kGKc / 4) * K::NB;
    float a = buf[a_off];
#pragma unroll
    for (int t = 0; t < TOT; ++t) {
      const int kq = t / K::NB, nb = t - kq * K::NB;
      const float b = buf[kq * 4 * K::PS2 + b_off + (nb / NBR) * 4 * K::RS + (nb % NBR) * 4];
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nb], 0, 0, 0);
      if (nb == K::NB - 1 && kq + 1 < kGKc / 4) a = buf[(kq + 1) * 4 * K::PS1 + a_off];
    }
  };
  (void)lds0;
#else
  auto mfma_phase = [&](const float* buf) {
    constexpr int PF = DTT_CORR_PF;
    static_assert(PF >= 3 && PF <= 14 && (kGKc / 4) * K::NB >= PF, "read-ahead ring");
    const unsigned base = lds0 + (unsigned)((buf - lds) * sizeof(float));
    const unsigned baddr = base + b_off * 4, aaddr = base + a_off * 4;
    float ring[PF], a[kGKc / 4];
    lds_read_async<0>(a[0], aaddr);
    if constexpr (kGKc / 4 > 1) lds_read_async<4 * K::PS1 * 4>(a[1], aaddr);
    static_assert(kGKc / 4 <= 2, "A operand prefetch covers two k-quads");
    corr_mfma_prime<K, NBR, PF, 0>(ring, baddr);
    lds_wait<PF>(a[0]);
    if constexpr (kGKc / 4 > 1) lds_wait<PF>(a[1]);
    __builtin_amdgcn_sched_barrier(0);
    corr_mfma_steps<K, NBR, PF, 0>(acc, ring, a, baddr);
  };
#endif
  // chunk ci lives in buffer ci & 1.  Top of a step: my DMA pieces have landed (vmcnt) and, past the barrier,
  // everybody's have -- and everybody is done reading the other buffer, which the next DMA may now overwrite.
  auto step = [&](float* cur, float* nxt, int ci) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tail_wg && ci == nch - 1) fix_tail(cur);
#if DTT_CORR_ABLATE != 11 && DTT_CORR_ABLATE != 13 && DTT_CORR_ABLATE != 15   // 11: no barriers, 12: no DMA, 13: neither (timing experiments only)
    __syncthreads();
#endif
#if DTT_CORR_ABLATE != 12 && DTT_CORR_ABLATE != 13 && DTT_CORR_ABLATE != 15
    if (ci + 1 < nch) issue(nxt, ci + 1 == nch - 1);
#endif
    if (wave_live) mfma_phase(cur);
  };
  float* buf0 = lds;
  float* buf1 = lds + K::BUF;
  issue(buf0, nch == 1);
  for (int ci = 0; ci < nch; ci += 2) {
    step(buf0, buf1, ci);
    if (ci + 1 < nch) step(buf1, buf0, ci + 1);
  }

  float* w = ws + ((((long)ks * nbatch + n) * ntiles + tile) * 4 + wave) * (long)(K::NB * 256);
#ifdef DTT_CORR_STAMP
  const unsigned long long stamp1 = __builtin_amdgcn_s_memrealtime();
#endif
#if DTT_CORR_ABLATE == 14 || DTT_CORR_ABLATE == 15   // 14: no partials store; 15: nor barriers / DMA
  if (g.C < 0)
#endif
#pragma unroll
  for (int nb = 0; nb < K::NB; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) w[(nb * 4 + r) * 64 + lane] = acc[nb][r];
#ifdef DTT_CORR_STAMP
  if (tid == 0) {
    unsigned long long* st = reinterpret_cast<unsigned long long*>(ws + (long)g.ksplit * nbatch * ntiles * 4 * K::NB * 256) +
                             4 * (long)blockIdx.x;
    st[0] = stamp0;
    st[1] = __builtin_amdgcn_s_memrealtime();
    st[2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    st[3] = (__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xF) | ((unsigned long long)item << 8) | ((stamp1 - stamp0) << 40);
  }
#endif
}

// grid: (tiles * 4 waves, NBR block rows, batch).  One workgroup sums the channel slices of one N-block row
// of one wave's fragment (slice order fixed -> deterministic), divides by nelems
// (correlation_cuda_kernel.cu:100 `reduce_sum / nelems`) and scatters the in-window entries to NCHW.
template <int NBR>
__global__ __launch_bounds__(kThreads) void corr_fwd_reduce(const float* __restrict__ ws, float* __restrict__ out,
                                                            long out_batch_stride, long out_ch_stride,
                                                            long out_px_stride, FastGeom g, float nelems) {
  using K = Cfg<NBR>;
  constexpr int ROW = NBR * 256;                 // floats of one N-block row: [nbx][reg][lane]
  __shared__ __attribute__((aligned(16))) float frag[ROW];
  const int ntiles = g.tiles_x * g.tiles_y;
  const int tile = blockIdx.x >> 2, wave = blockIdx.x & 3, nby = blockIdx.y, n = blockIdx.z;
  const int nbatch = gridDim.z;
  const int tid = threadIdx.x;
  constexpr int FR = K::NB * 256;
  const long slice_stride = (long)nbatch * ntiles * 4 * FR;
  const float4* src = reinterpret_cast<const float4*>(ws + (((long)n * ntiles + tile) * 4 + wave) * (long)FR + nby * ROW);
  const long slice4 = slice_stride / 4;
  for (int i = tid; i < ROW / 4; i += kThreads) {
    float4 s = src[i];
    for (int k = 1; k < g.ksplit; ++k) {
      const float4 v = src[i + k * slice4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(frag)[i] = s;
  }
  __syncthreads();
  const int ty0 = (tile / g.tiles_x) * kTile + (wave >> 1) * 4, tx0 = (tile % g.tiles_x) * kTile + (wave & 1) * 4;
  float* o = out + (long)n * out_batch_stride;
  // outputs whose halo row hr = tj + iy falls in this block row: tj in [4*nby - 3, 4*nby + 3]
  const int tj_lo = max(0, 4 * nby - 3), tj_hi = min(g.D - 1, 4 * nby + 3);
  const int total = (tj_hi - tj_lo + 1) * g.D * 16;
  // thread order follows the output layout: NCHW (channel stride > 1) walks the 16 pixels of the block fastest,
  // position-major output (channel stride 1: the displacements of a pixel are contiguous) the displacements
  const int nq = (tj_hi - tj_lo + 1) * g.D;
  for (int idx = tid; idx < total; idx += kThreads) {
    const int p = out_ch_stride == 1 ? idx / nq : idx & 15, q = out_ch_stride == 1 ? idx - p * nq : idx >> 4;
    const int tj = tj_lo + q / g.D, ti = q % g.D;  // already offset by +R
    const int iy = p >> 2, ix = p & 3;
    const int hr = tj + iy;
    if ((hr >> 2) != nby) continue;
    const int y = ty0 + iy, x = tx0 + ix;
    if (y >= g.oh || x >= g.ow) continue;
    const int hc = ti + ix;
    const int lane = iy * 16 + (hr & 3) * 4 + (hc & 3);
    // zero padding, applied on the output side (the LDS-DMA kernel stages unmasked pixels)
    const int py = g.origin + y * g.s, px = g.origin + x * g.s;
    const int tjg = tj + g.qy + g.Rfull - g.R, tig = ti + g.qx + g.Rfull - g.R;   // index in the full window
    const int qy = py + (tjg - g.Rfull) * g.s, qx = px + (tig - g.Rfull) * g.s;
    const bool in_image = py >= 0 && py < g.H && px >= 0 && px < g.W && qy >= 0 && qy < g.H && qx >= 0 && qx < g.W;
    o[(long)(tjg * g.Dfull + tig) * out_ch_stride + ((long)y * g.ow + x) * out_px_stride] = in_image ? frag[((hc >> 2) * 4 + ix) * 64 + lane] / nelems : 0.f;
  }
}

// Generic forward (any kernel_size / strides): one wave per output pixel, lanes over channels, wave
// reduction per displacement.  Slow path, kept for API completeness (correlation.py:5-13 defaults).
__global__ __launch_bounds__(64) void corr_fwd_generic(const float* __restrict__ in1, const float* __restrict__ in2,
                                                       float* __restrict__ out, long out_batch_stride,
                                                       long out_ch_stride, long out_px_stride, int C, int H,
                                                       int W, int oc, int oh, int ow, int pad, int ksize, int maxd,
                                                       int s1, int s2) {
  const int n = blockIdx.z, by = blockIdx.y, bx = blockIdx.x, lane = threadIdx.x;
  const int krad = (ksize - 1) / 2, drad = maxd / s2, dsize = 2 * drad + 1;
  const int pH = H + 2 * pad, pW = W + 2 * pad;
  const int y1 = by * s1 + maxd + krad, x1 = bx * s1 + maxd + krad;  // padded coordinates (.cu:48-49)
  const long plane = (long)H * W;
  const float* p1 = in1 + (long)n * C * plane;
  const float* p2 = in2 + (long)n * C * plane;
  const float nelems = (float)(ksize * ksize * C);
  for (int tj = -drad; tj <= drad; ++tj)
    for (int ti = -drad; ti <= drad; ++ti) {
      const int x2 = x1 + ti * s2, y2 = y1 + tj * s2;
      float acc = 0.f;
      for (int j = -krad; j <= krad; ++j)
        for (int i = -krad; i <= krad; ++i) {
          const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
          const bool pa = ya >= 0 && ya < pH && xa >= 0 && xa < pW, pb = yb >= 0 && yb < pH && xb >= 0 && xb < pW;
          const int ua = ya - pad, va = xa - pad, ub = yb - pad, vb = xb - pad;
          const bool ina = pa && ua >= 0 && ua < H && va >= 0 && va < W;
          const bool inb = pb && ub >= 0 && ub < H && vb >= 0 && vb < W;
          if (!(ina && inb)) continue;
          for (int ch = lane; ch < C; ch += 64) acc += p1[ch * plane + ua * W + va] * p2[ch * plane + ub * W + vb];
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0) {
        const int tc = (tj + drad) * dsize + (ti + drad);
        out[(long)n * out_batch_stride + (long)tc * out_ch_stride + ((long)by * ow + bx) * out_px_stride] = acc / nelems;
      }
    }
}

// Backward, any kernel_size / strides.  One thread per input element; gradients are the mathematically exact adjoint
// of the forward (see dtt_hip.h for where the reference's own backward departs from its forward).  With the forward
//   out[n,tc,oy,ox] = 1/(k*k*C) * sum_{c, 0<=j,i<k} in1pad[n,c,oy*s1+d+j, ox*s1+d+i] * in2pad[n,c,oy*s1+d+j+j2, ox*s1+d+i+i2]
// (padded coordinates, (j2, i2) = displacement tc, d = max_displacement) a padded pixel (yp, xp) of frame 1 is touched by
// the outputs with oy*s1 in [yp-d-(k-1), yp-d], and one of frame 2 by those with oy*s1 in [yp-d-j2-(k-1), yp-d-j2]:
//   gradInput1[n,c,y,x] = 1/(k*k*C) * sum_tc in2pad[n,c,yp+j2,xp+i2] * sum_{covering (oy,ox)} gradOut[n,tc,oy,ox]
//   gradInput2[n,c,y,x] = 1/(k*k*C) * sum_tc in1pad[n,c,yp-j2,xp-i2] * sum_{covering (oy,ox)} gradOut[n,tc,oy,ox]
__device__ __forceinline__ int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__device__ __forceinline__ int ceildiv(int a, int b) { return -floordiv(-a, b); }

__global__ void corr_bwd_simple(const float* __restrict__ gout, const float* __restrict__ in1,
                                const float* __restrict__ in2, float* __restrict__ g1, float* __restrict__ g2, int B,
                                int C, int H, int W, int oc, int oh, int ow, int pad, int ksize, int maxd, int s1, int s2) {
  const long total = (long)B * C * H * W;
  const int drad = maxd / s2, dsize = 2 * drad + 1, kext = ksize - 1;
  const float nelems = (float)(ksize * ksize * C);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int x = idx % W, y = (idx / W) % H, c = (idx / ((long)W * H)) % C, n = idx / ((long)W * H * C);
    const float* go = gout + (long)n * oc * oh * ow;
    const float* f1 = in1 + ((long)n * C + c) * H * W;
    const float* f2 = in2 + ((long)n * C + c) * H * W;
    const int yp = y + pad, xp = x + pad;
    // ---- input1
    float a1 = 0.f;
    {
      const int oy0 = max(ceildiv(yp - maxd - kext, s1), 0), oy1 = min(floordiv(yp - maxd, s1), oh - 1);
      const int ox0 = max(ceildiv(xp - maxd - kext, s1), 0), ox1 = min(floordiv(xp - maxd, s1), ow - 1);
      if (oy0 <= oy1 && ox0 <= ox1) {
        for (int tc = 0; tc < oc; ++tc) {
          const int i2 = (tc % dsize - drad) * s2, j2 = (tc / dsize - drad) * s2;
          const int yb = y + j2, xb = x + i2;
          if (yb < 0 || yb >= H || xb < 0 || xb >= W) continue;
          float gs = 0.f;
          for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) gs += go[((long)tc * oh + oy) * ow + ox];
          a1 += gs * f2[yb * W + xb];
        }
      }
    }
    // ---- input2
    float a2 = 0.f;
    for (int tc = 0; tc < oc; ++tc) {
      const int i2 = (tc % dsize - drad) * s2, j2 = (tc / dsize - drad) * s2;
      const int ya = y - j2, xa = x - i2;
      if (ya < 0 || ya >= H || xa < 0 || xa >= W) continue;
      const int oy0 = max(ceildiv(yp - maxd - j2 - kext, s1), 0), oy1 = min(floordiv(yp - maxd - j2, s1), oh - 1);
      const int ox0 = max(ceildiv(xp - maxd - i2 - kext, s1), 0), ox1 = min(floordiv(xp - maxd - i2, s1), ow - 1);
      float gs = 0.f;
      for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) gs += go[((long)tc * oh + oy) * ow + ox];
      a2 += gs * f1[ya * W + xa];
    }
    g1[idx] = a1 / nelems;
    g2[idx] = a2 / nelems;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward on the matrix cores (kernel_size 1, stride1 == stride2).  With G[p, q] = gradOut[tc(q - p), p] the
// banded gradient matrix (p: output pixel, q: displaced pixel, zero outside the window),
//     gradInput1[c, p] = 1/C * sum_q f2[c, q] * G[p, q]         gradInput2[c, q] = 1/C * sum_p f1[c, p] * G[p, q]
// i.e. for a 4x4 block of target pixels the reduction runs over the (4+2R)^2 halo of the OTHER frame: a
// [16 channels x 4 halo pixels] x [4 halo pixels x 16 targets] MFMA per step.  The band G does not depend on the
// channel, so every wave gathers its NB*4 band words from gradOut ONCE into registers (the MFMA B operand)
// and then streams channel chunks of the other frame through LDS (A operand) -- no split-K, no atomics, each
// gradient element is written exactly once.  WRT2 selects which gradient (and hence which gather rule).
// NHWC: `other` and `grad` are channels-last (n, y, x, c) -- the training trunk's layout: a halo pixel's 16-channel chunk is one
// 64-byte piece whatever the lattice stride (so conv3 gets the pipelined 16-byte staging too), LDS holds the halo pixel-major
// ([pixel][16 channels]: the A read of 16 channels x 4 neighbouring pixels is 64 consecutive floats, conflict free) and a lane
// stores its four channels of a target pixel as one float4.  Same MFMA sequence, so the gradients are bit-identical to the
// NCHW instantiation's.
template <int NBR, bool WRT2, int MINW, bool NHWC = false>
__global__ __launch_bounds__(kThreads, MINW) void corr_bwd_mfma(const float* __restrict__ gout,
                                                                const float* __restrict__ other,
                                                                float* __restrict__ grad, FastGeom g, int cgroups,
                                                                float nelems, int toff_y, int toff_x, int tiles_y,
                                                                int tiles_x) {
  using K = Cfg<NBR>;
  constexpr int KB = 16;                               // channels per chunk = MFMA M
  constexpr int G4 = K::HR / 4, PL4 = K::HR * G4;
  constexpr int N4 = (KB * PL4 + kThreads - 1) / kThreads;
  constexpr int NS = (KB * K::HR * K::HR + kThreads - 1) / kThreads;
  constexpr int PSB = ((K::HR * K::HRS + 31) / 32) * 32 + 2;  // plane stride = 2 (mod 32): A reads conflict free
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [KB][PSB]

  // targets are tiled on the lattice starting at (toff_y, toff_x): the output pixels for gradInput1, the
  // displaced pixels that fall inside the image for gradInput2
  const int ntiles = tiles_x * tiles_y;
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % ntiles;
  const int cg = (item / ntiles) % cgroups, n = item / (ntiles * cgroups);
  const int ty0 = toff_y + (tile / tiles_x) * kTile, tx0 = toff_x + (tile % tiles_x) * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int my = wave >> 1, mx = wave & 1;
  const int plane = g.H * g.W;
  const int c_per = ((g.C + cgroups - 1) / cgroups + KB - 1) / KB * KB;
  const int c_begin = cg * c_per, c_end = min(g.C, c_begin + c_per);

  // ---- band words: B[k = lane>>4][j = lane&15] for step t = nb*4 + jy  ->  halo pixel (nby*4+jy, nbx*4+k) of this
  //      wave's halo, target pixel j = (ty, tx) of this wave's 4x4 block
  const int kx = lane >> 4, jt = lane & 15, tyi = jt >> 2, txi = jt & 3;
  float band[K::NB * 4];
  {
    const float* go = gout + (long)n * g.oc * g.oh * g.ow;
    const int ty = ty0 + my * 4 + tyi, tx = tx0 + mx * 4 + txi;  // target lattice pixel
#pragma unroll
    for (int t = 0; t < K::NB * 4; ++t) {
      const int nb = t >> 2, jy = t & 3;
      const int hy = (nb / NBR) * 4 + jy, hx = (nb % NBR) * 4 + kx;  // halo pixel relative to the wave's halo origin
      const int oy = ty0 + my * 4 - g.R + hy, ox = tx0 + mx * 4 - g.R + hx;  // lattice position of the halo pixel
      int tj, ti, py, px;
      if (!WRT2) { tj = hy - tyi; ti = hx - txi; py = ty; px = tx; }           // q = halo, p = target
      else       { tj = 2 * g.R - (hy - tyi); ti = 2 * g.R - (hx - txi); py = oy; px = ox; }  // p = halo, q = target
      const bool ok = tj >= 0 && tj < g.D && ti >= 0 && ti < g.D && py >= 0 && py < g.oh && px >= 0 && px < g.ow;
      band[t] = ok ? go[((long)(tj * g.D + ti) * g.oh + py) * g.ow + px] : 0.f;
    }
  }

  // ---- staging descriptors for the halo of the other frame (same geometry as the forward kernel)
  const bool vec4 = g.s == 1 && g.W >= 4;
  int goff4[N4], loff4[N4], dl4[N4];
  unsigned vb4[N4];
  bool fix = false;
#pragma unroll
  for (int i = 0; i < N4; ++i) {
    const int f = tid + i * kThreads;
    const bool live = f < KB * PL4;
    const int cc = live ? f / PL4 : 0, r = live ? f - cc * PL4 : 0;
    const int hr = r / G4, q = r - hr * G4;
    const int gy = g.origin + ty0 - g.R + hr, start = g.origin + tx0 - g.R + 4 * q;  // vec4 path: stride 1
    const bool row_ok = live && gy >= 0 && gy < g.H;
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) bits |= (row_ok && start + j >= 0 && start + j < g.W) ? (1u << j) : 0u;
    const int ld = min(max(start, 0), g.W - 4);
    vb4[i] = bits;
    dl4[i] = bits ? start - ld : 0;
    goff4[i] = bits ? cc * plane + gy * g.W + ld : 0;
    loff4[i] = live ? cc * PSB + hr * K::HRS + 4 * q : KB * PSB;
    fix |= dl4[i] != 0;
  }
  const bool wg_fix = __syncthreads_or(fix);

  const float* src = other + (long)n * g.C * plane;
  float* dst = grad + (long)n * g.C * plane;
  const int a_off = (lane & 15) * PSB + (my * 4) * K::HRS + mx * 4 + kx;  // A[i = channel][k = halo column kx]

  auto load16 = [](const float* p) -> f32x4 { f32x4 v; __builtin_memcpy(&v, p, 16); return v; };
  f32x4 st[N4];

  auto stage_vec = [&](int c0, bool issue, bool write) {
    if (issue) {
#pragma unroll
      for (int i = 0; i < N4; ++i) st[i] = load16(src + (long)c0 * plane + goff4[i]);
    }
    if (write) {
#pragma unroll
      for (int i = 0; i < N4; ++i) {
        f32x4 o = st[i];
        if (wg_fix) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = j + dl4[i];
            float x = st[i][0];
            x = k == 1 ? st[i][1] : x;
            x = k == 2 ? st[i][2] : x;
            x = k == 3 ? st[i][3] : x;
            o[j] = x;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = __uint_as_float(__float_as_uint(o[j]) & (unsigned)(-(int)((vb4[i] >> j) & 1u)));
        *reinterpret_cast<f32x4*>(lds + loff4[i]) = o;
      }
    }
  };
  // scalar staging for strided lattices (conv3: stride 2), not pipelined
  auto stage_scalar = [&](int c0) {
#pragma unroll 4
    for (int e = 0; e < NS; ++e) {
      const int f = tid + e * kThreads;
      if (f < KB * K::HR * K::HR) {
        const int cc = f / (K::HR * K::HR), r = f - cc * (K::HR * K::HR);
        const int hr = r / K::HR, hc = r - hr * K::HR;
        const int gy = g.origin + (ty0 - g.R + hr) * g.s, gx = g.origin + (tx0 - g.R + hc) * g.s;
        const bool ok = gy >= 0 && gy < g.H && gx >= 0 && gx < g.W && (c0 + cc) < g.C;
        lds[cc * PSB + hr * K::HRS + hc] = ok ? src[(long)(c0 + cc) * plane + gy * g.W + gx] : 0.f;
      }
    }
  };

  // channels-last staging: piece f = (halo pixel, 16-byte quarter of its 64-byte chunk row)
  constexpr int NP = NHWC ? (K::HR * K::HR * 4 + kThreads - 1) / kThreads : 1;
  long goffp[NP];
  int loffp[NP];
  bool okp[NP];
  f32x4 stp[NP];
  if constexpr (NHWC) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int f = tid + i * kThreads;
      const bool live = f < K::HR * K::HR * 4;
      const int pi = live ? f >> 2 : 0, piece = f & 3;
      const int hr = pi / K::HR, hc = pi - hr * K::HR;
      const int gy = g.origin + (ty0 - g.R + hr) * g.s, gx = g.origin + (tx0 - g.R + hc) * g.s;
      okp[i] = live && gy >= 0 && gy < g.H && gx >= 0 && gx < g.W;
      goffp[i] = okp[i] ? ((long)gy * g.W + gx) * g.C + piece * 4 : 0;
      loffp[i] = live ? pi * KB + piece * 4 : K::HR * K::HR * KB;   // (idle lanes: the spare slot behind the image)
    }
  }
  auto stage_nhwc = [&](int c0, bool issue, bool write) {
    if (issue) {
#pragma unroll
      for (int i = 0; i < NP; ++i) stp[i] = load16(src + goffp[i] + c0);
    }
    if (write) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        f32x4 o = stp[i];
        if (!okp[i]) o = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(lds + loffp[i]) = o;
      }
    }
  };
  const int a_off_p = ((my * 4) * K::HR + mx * 4 + kx) * KB + (lane & 15);   // A[i = channel][k = halo column kx], pixel-major

  // target pixel of this lane's D column (j = lane & 15) and validity of the write
  const int wy = g.origin + (ty0 + my * 4 + tyi) * g.s, wx = g.origin + (tx0 + mx * 4 + txi) * g.s;
  const bool w_ok = wy >= 0 && wy < g.H && wx >= 0 && wx < g.W;

  if (c_begin >= c_end) return;   // more channel groups than 16-channel chunks (small C): nothing to do, and no staging --
                                  // the prologue below would read past the end of the tensor (found by tools/fuzz_ops.py)
  if constexpr (NHWC) { stage_nhwc(c_begin, true, true); }
  else if (vec4) { stage_vec(c_begin, true, true); } else { stage_scalar(c_begin); }
  __syncthreads();
  for (int c0 = c_begin; c0 < c_end; c0 += KB) {
    const bool more = c0 + KB < c_end;
    if constexpr (NHWC) { if (more) stage_nhwc(c0 + KB, true, false); }
    else if (vec4 && more) stage_vec(c0 + KB, true, false);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < K::NB * 4; t += 2) {
      const int nb0 = t >> 2, jy0 = t & 3, nb1 = (t + 1) >> 2, jy1 = (t + 1) & 3;
      float a0, a1;
      if constexpr (NHWC) {
        a0 = lds[a_off_p + (((nb0 / NBR) * 4 + jy0) * K::HR + (nb0 % NBR) * 4) * KB];
        a1 = lds[a_off_p + (((nb1 / NBR) * 4 + jy1) * K::HR + (nb1 % NBR) * 4) * KB];
      } else {
        a0 = lds[a_off + ((nb0 / NBR) * 4 + jy0) * K::HRS + (nb0 % NBR) * 4];
        a1 = lds[a_off + ((nb1 / NBR) * 4 + jy1) * K::HRS + (nb1 % NBR) * 4];
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, band[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, band[t + 1], acc1, 0, 0, 0);
    }
    // D[i = channel (lane>>4)*4 + reg][j = target pixel]
    if (w_ok) {
      if constexpr (NHWC) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (acc0[r] + acc1[r]) / nelems;
        *reinterpret_cast<f32x4*>(dst + ((long)wy * g.W + wx) * g.C + c0 + (lane >> 4) * 4) = o;   // (C % 16 == 0: no channel tail)
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = c0 + (lane >> 4) * 4 + r;
          if (c < c_end) dst[(long)c * plane + wy * g.W + wx] = (acc0[r] + acc1[r]) / nelems;
        }
      }
    }
    __syncthreads();
    if (more) {
      if constexpr (NHWC) stage_nhwc(0, false, true);
      else if (vec4) stage_vec(0, false, true); else stage_scalar(c0 + KB);
      __syncthreads();
    }
  }
}

template <int NBR, bool NHWC>
size_t bwd_lds_bytes() {
  using K = Cfg<NBR>;
  if (NHWC) return (size_t)(K::HR * K::HR * 16 + 4) * sizeof(float);
  return (size_t)(16 * (((K::HR * K::HRS + 31) / 32) * 32 + 2) + 4) * sizeof(float);
}

template <int NBR, int MINW, bool NHWC>
int launch_bwd(const float* gout, const float* in1, const float* in2, float* g1, float* g2, const FastGeom& g,
               int batch, hipStream_t stream) {
  const size_t lds = bwd_lds_bytes<NBR, NHWC>();
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();   // the attribute is per device, not per process
  if (!attr) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_bwd_mfma<NBR, false, MINW, NHWC>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_bwd_mfma<NBR, true, MINW, NHWC>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DTT_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "correlation backward: cannot raise dynamic LDS limit");
    attr = true;
  }
  const size_t bytes = (size_t)batch * g.C * g.H * g.W * sizeof(float);
  // pixels that no lattice point maps to (stride > 1, pad != displacement) keep a zero gradient; on the dense lattice of
  // conv4 / conv5 (stride 1, pad == displacement: output pixel = input pixel) both kernels write every element themselves
  const bool every_pixel_is_a_target = g.s == 1 && g.origin == 0 && g.oh == g.H && g.ow == g.W;
  if (!every_pixel_is_a_target)
    DTT_REQUIRE(hipMemsetAsync(g1, 0, bytes, stream) == hipSuccess && hipMemsetAsync(g2, 0, bytes, stream) == hipSuccess,
                "correlation backward: memset failed");
  auto cdiv_floor = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
  // channel groups of the gradient kernels: MINW workgroups are resident per CU, so tiles x groups x batch is kept within ONE
  // round of resident workgroups (the forward's 720-workgroup split is 1.4 rounds here: a second, 40 %-full round as long as
  // the first)
  static const int ks_env = getenv("DTT_CORR_BWD_KSPLIT") ? atoi(getenv("DTT_CORR_BWD_KSPLIT")) : 0;   // developer A/B switch
  auto groups_for = [&](int tiles) {
    int ks = ks_env > 0 ? ks_env : (MINW * dtt_device_cus()) / (tiles * batch > 0 ? tiles * batch : 1);
    const int max_ks = (g.C + 15) / 16;
    ks = ks > max_ks ? max_ks : ks;
    return ks < 1 ? 1 : ks;
  };
  // gradInput1: targets = output pixels (lattice 0 .. oh-1)
  {
    const int ty = g.tiles_y, tx = g.tiles_x, ks = groups_for(ty * tx);
    hipLaunchKernelGGL((corr_bwd_mfma<NBR, false, MINW, NHWC>), dim3(ty * tx * ks * batch), dim3(kThreads), lds, stream,
                       gout, in2, g1, g, ks, (float)g.C, 0, 0, ty, tx);
    DTT_CHECK_LAUNCH("corr_bwd_mfma<input1>");
  }
  // gradInput2: targets = displaced pixels q = p + disp that fall inside the image
  {
    const int qlo_y = -cdiv_floor(g.origin, g.s), qlo_x = qlo_y;               // smallest lattice index with pixel >= 0
    const int qhi_y = cdiv_floor(g.H - 1 - g.origin, g.s), qhi_x = cdiv_floor(g.W - 1 - g.origin, g.s);
    const int lo_y = qlo_y > -g.R ? qlo_y : -g.R, lo_x = qlo_x > -g.R ? qlo_x : -g.R;
    const int hi_y = qhi_y < g.oh - 1 + g.R ? qhi_y : g.oh - 1 + g.R, hi_x = qhi_x < g.ow - 1 + g.R ? qhi_x : g.ow - 1 + g.R;
    if (hi_y >= lo_y && hi_x >= lo_x) {
      const int ty = (hi_y - lo_y + kTile) / kTile, tx = (hi_x - lo_x + kTile) / kTile, ks = groups_for(ty * tx);
      hipLaunchKernelGGL((corr_bwd_mfma<NBR, true, MINW, NHWC>), dim3(ty * tx * ks * batch), dim3(kThreads), lds,
                         stream, gout, in1, g2, g, ks, (float)g.C, lo_y, lo_x, ty, tx);
      DTT_CHECK_LAUNCH("corr_bwd_mfma<input2>");
    }
  }
  return 1;
}

bool fast_path(int ksize, int s1, int s2, int maxd, int* nbr) {
  if (ksize != 1 || s1 != s2) return false;
  const int R = maxd / s2;
  if (R < 1 || R > 16) return false;
  *nbr = R <= 4 ? 3 : (R <= 8 ? 5 : 9);
  return true;
}

FastGeom make_geom(int batch, int C, int H, int W, int oc, int oh, int ow, int pad, int maxd, int s) {
  FastGeom g;
  g.C = C; g.H = H; g.W = W; g.oc = oc; g.oh = oh; g.ow = ow; g.s = s;
  g.origin = maxd - pad;
  g.R = maxd / s;
  g.D = 2 * g.R + 1;
  g.qy = g.qx = 0; g.Rfull = g.R; g.Dfull = g.D;
  g.tiles_x = (ow + kTile - 1) / kTile;
  g.tiles_y = (oh + kTile - 1) / kTile;
  // Channel slices: about 3 workgroups per CU (more only inflates the split-K partials), and slices x batch a
  // multiple of the 8 XCDs so that every slice's halo re-reads stay inside one XCD's L2 (measured: 8 slices at
  // B=2 beat 6, 10, 12 and 16 by 6-25 %).
  const int tiles = g.tiles_x * g.tiles_y;
#ifdef DTT_CORR_KSPLIT
  int ks = DTT_CORR_KSPLIT;
#else
  int gcd = 8, bb = batch;
  while (bb) { const int t = gcd % bb; gcd = bb; bb = t; }
  const int step = 8 / gcd;
  int ks = (int)(720.0 / ((double)tiles * batch) / step + 0.5) * step;
  if (ks < step) ks = step;
#endif
  const int max_ks = (C + kKc - 1) / kKc;
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  int cps = (C + ks - 1) / ks;
  cps = ((cps + kKc - 1) / kKc) * kKc;
  g.c_per_split = cps;
  g.ksplit = (C + cps - 1) / cps;
  return g;
}

template <int NBR>
size_t ws_bytes(const FastGeom& g, int batch) {
  return (size_t)g.ksplit * batch * g.tiles_x * g.tiles_y * 4 * Cfg<NBR>::NB * 256 * sizeof(float);
}

template <int NBR, bool PIPE, int MINW, bool VEC4>
int launch_fast(float* output, long out_batch_stride, long out_ch_stride, long out_px_stride, const float* in1, const float* in2, void* workspace,
                size_t workspace_bytes, const FastGeom& g, int batch, hipStream_t stream) {
  using K = Cfg<NBR>;
  const size_t need = ws_bytes<NBR>(g, batch);
  DTT_REQUIRE(workspace && workspace_bytes >= need, "correlation forward: workspace too small (%zu < %zu)",
              workspace_bytes, need);
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();   // the attribute is per device, not per process
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_fwd_mfma<NBR, PIPE, MINW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS);
    hipError_t e2 = hipSuccess;
    if constexpr (VEC4)
      e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_fwd_mfma_v4<NBR, MINW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS);
    DTT_REQUIRE(e == hipSuccess && e2 == hipSuccess, "correlation: cannot raise dynamic LDS limit");
    attr = true;
  }
  const int ntiles = g.tiles_x * g.tiles_y;
  dtt_prof_begin("corr_fwd_op", stream);     // the whole op: banded product + slice reduction
  dtt_prof_begin("corr_fwd_mfma", stream);
  bool launched = false;
#ifndef DTT_CORR_NO_GLDS
  // (R <= 8 only: with the 81 accumulators of the R <= 16 instantiation the LDS-DMA kernel spills and runs 3x slower
  // than the register-staged one)
  if constexpr (NBR <= 5)   // (compile time: the 81-accumulator LDS-DMA instantiation is not built at all)
  if (g.s == 1 && g.C % kGKc == 0 && g.c_per_split % kGKc == 0 && g.W >= 4 && (((g.origin - g.R) % 4) + 4) % 4 == 0) {
    using G = GCfg<NBR>;
    static DttDeviceOnce gattr_once;
  bool& gattr = gattr_once.here();   // the attribute is per device, not per process
    if (!gattr) {
      hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_fwd_glds<NBR, MINW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
      DTT_REQUIRE(e3 == hipSuccess, "correlation: cannot raise dynamic LDS limit (LDS-DMA kernel)");
      gattr = true;
    }
    hipLaunchKernelGGL((corr_fwd_glds<NBR, MINW>), dim3(ntiles * g.ksplit * batch), dim3(kThreads), G::LDS, stream,
                       in1, in2, static_cast<float*>(workspace), g);
    launched = true;
  }
#endif
  if (!launched) {
    DTT_REQUIRE(g.qy == 0 && g.qx == 0, "correlation: window decomposition needs the LDS-DMA kernel");
    if constexpr (VEC4) {
      if (g.s == 1 && g.C % kKc == 0 && g.W >= 4) {
        hipLaunchKernelGGL((corr_fwd_mfma_v4<NBR, MINW>), dim3(ntiles * g.ksplit * batch), dim3(kThreads), K::LDS, stream,
                           in1, in2, static_cast<float*>(workspace), g);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((corr_fwd_mfma<NBR, PIPE, MINW>), dim3(ntiles * g.ksplit * batch), dim3(kThreads), K::LDS, stream,
                         in1, in2, static_cast<float*>(workspace), g);
  }
  dtt_prof_end("corr_fwd_mfma", stream);
  DTT_CHECK_LAUNCH("corr_fwd_mfma");
  dtt_prof_begin("corr_fwd_reduce", stream);
  hipLaunchKernelGGL(corr_fwd_reduce<NBR>, dim3(ntiles * 4, NBR, batch), dim3(kThreads), 0, stream,
                     static_cast<const float*>(workspace), output, out_batch_stride, out_ch_stride, out_px_stride, g, (float)g.C);
  dtt_prof_end("corr_fwd_reduce", stream);
  dtt_prof_end("corr_fwd_op", stream);
  DTT_CHECK_LAUNCH("corr_fwd_reduce");
  return 1;
}

}  // namespace

// correlation_cuda.c:19-34
extern "C" int dtt_correlation_output_shape(int ic, int ih, int iw, int pad_size, int kernel_size,
                                            int max_displacement, int stride1, int stride2, int* oc, int* oh,
                                            int* ow) {
  DTT_REQUIRE(oc && oh && ow, "correlation: null output pointer");
  DTT_REQUIRE(ic > 0 && ih > 0 && iw > 0, "correlation: bad input shape");
  DTT_REQUIRE(stride1 > 0 && stride2 > 0 && kernel_size > 0 && (kernel_size & 1) && pad_size >= 0 &&
                  max_displacement >= 0,
              "correlation: bad parameters (kernel_size must be odd and > 0, strides > 0)");
  const int kernel_radius = (kernel_size - 1) / 2;
  const int border_radius = kernel_radius + max_displacement;
  const int pH = ih + 2 * pad_size, pW = iw + 2 * pad_size;
  const int r = max_displacement / stride2;
  *oc = (2 * r + 1) * (2 * r + 1);
  *oh = (int)ceilf((float)(pH - 2 * border_radius) / (float)stride1);
  *ow = (int)ceilf((float)(pW - 2 * border_radius) / (float)stride1);
  DTT_REQUIRE(*oh > 0 && *ow > 0, "correlation: empty output (%d x %d)", *oh, *ow);
  return 1;
}

extern "C" size_t dtt_correlation_forward_workspace_bytes(int batch, int ic, int ih, int iw, int pad_size,
                                                          int kernel_size, int max_displacement, int stride1,
                                                          int stride2) {
  int oc, oh, ow, nbr;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow))
    return 0;
  if (!fast_path(kernel_size, stride1, stride2, max_displacement, &nbr)) return 0;
  const FastGeom g = make_geom(batch, ic, ih, iw, oc, oh, ow, pad_size, max_displacement, stride1);
  return nbr == 3 ? ws_bytes<3>(g, batch) : (nbr == 5 ? ws_bytes<5>(g, batch) : ws_bytes<9>(g, batch));
}

extern "C" int dtt_correlation_forward_strided(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                               long out_ch_stride, long out_px_stride, const float* input1, int ic,
                                               int ih, int iw, const float* input2, void* workspace,
                                               size_t workspace_bytes, int pad_size, int kernel_size,
                                               int max_displacement, int stride1, int stride2, int corr_type_multiply,
                                               void* stream_);

extern "C" int dtt_correlation_forward(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                       const float* input1, int ic, int ih, int iw, const float* input2,
                                       void* workspace, size_t workspace_bytes, int pad_size, int kernel_size,
                                       int max_displacement, int stride1, int stride2, int corr_type_multiply,
                                       void* stream_) {
  DTT_REQUIRE(out_batch_stride >= (long)oc * oh * ow, "correlation forward: out_batch_stride too small");
  return dtt_correlation_forward_strided(output, ob, oc, oh, ow, out_batch_stride, (long)oh * ow, 1, input1, ic, ih, iw,
                                         input2, workspace, workspace_bytes, pad_size, kernel_size, max_displacement,
                                         stride1, stride2, corr_type_multiply, stream_);
}

// Element (n, d, y, x) of the output lives at output[n * out_batch_stride + d * out_ch_stride + (y * ow + x) *
// out_px_stride]: (oh*ow, 1) is the reference's NCHW tensor (or a channel slice of a bigger one), (1, ld) a column
// block of a position-major (pixels, ld) matrix -- the layout the tracking head's GEMM consumes.
extern "C" int dtt_correlation_forward_strided(float* output, int ob, int oc, int oh, int ow, long out_batch_stride,
                                               long out_ch_stride, long out_px_stride, const float* input1, int ic,
                                               int ih, int iw, const float* input2, void* workspace,
                                               size_t workspace_bytes, int pad_size, int kernel_size,
                                               int max_displacement, int stride1, int stride2, int corr_type_multiply,
                                               void* stream_) {
  (void)corr_type_multiply;  // accepted and ignored, as in the reference kernels
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(output && input1 && input2, "correlation forward: null pointer");
  int eoc, eoh, eow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &eoc, &eoh, &eow))
    return 0;
  DTT_REQUIRE(ob > 0 && oc == eoc && oh == eoh && ow == eow,
              "correlation forward: output is (%d,%d,%d,%d), expected (B,%d,%d,%d)", ob, oc, oh, ow, eoc, eoh, eow);
  DTT_REQUIRE(out_ch_stride > 0 && out_px_stride > 0 && out_batch_stride > 0, "correlation forward: bad output strides");
  int nbr;
  if (fast_path(kernel_size, stride1, stride2, max_displacement, &nbr)) {
    const FastGeom g = make_geom(ob, ic, ih, iw, oc, oh, ow, pad_size, max_displacement, stride1);
    if (nbr == 3) return launch_fast<3, true, DTT_CORR_MINW, true>(output, out_batch_stride, out_ch_stride, out_px_stride, input1, input2, workspace, workspace_bytes, g, ob, stream);
    if (nbr == 5) return launch_fast<5, true, DTT_CORR_MINW, true>(output, out_batch_stride, out_ch_stride, out_px_stride, input1, input2, workspace, workspace_bytes, g, ob, stream);
#ifndef DTT_CORR_NO_QUADRANTS
    // 8 < R <= 16 (BASELINE config 5: d = 16): the 81-accumulator instantiation runs one wave per SIMD.  The window is
    // instead covered by four (2*8+1)^2 sub-windows centred at (+-(R-8), +-(R-8)) -- the same arithmetic per output, so
    // the overlapping rows / columns are simply written twice with identical values -- each one a launch of the R <= 8
    // LDS-DMA kernel on frame t+tau shifted by the centre.  Needs 16-byte aligned halo columns: R a multiple of 4.
    if (g.s == 1 && g.R % 4 == 0 && ic % kGKc == 0 && iw >= 4 && (((g.origin - 8) % 4) + 4) % 4 == 0) {
      FastGeom q = g;
      q.R = 8; q.D = 17;
      q.Rfull = g.R; q.Dfull = g.D;
      const int c = g.R - 8;
      for (int sy = -1; sy <= 1; sy += 2)
        for (int sx = -1; sx <= 1; sx += 2) {
          q.qy = sy * c; q.qx = sx * c;
#ifdef DTT_CORR_QDEBUG
          q.qy = DTT_CORR_QDEBUG_Y; q.qx = DTT_CORR_QDEBUG_X;   // one fixed window (debug)
#endif
          if (!launch_fast<5, true, DTT_CORR_MINW, true>(output, out_batch_stride, out_ch_stride, out_px_stride, input1, input2,
                                                         workspace, workspace_bytes, q, ob, stream))
            return 0;
        }
      return 1;
    }
#endif
    return launch_fast<9, false, 1, false>(output, out_batch_stride, out_ch_stride, out_px_stride, input1, input2, workspace, workspace_bytes, g, ob, stream);
  }
  hipLaunchKernelGGL(corr_fwd_generic, dim3(ow, oh, ob), dim3(64), 0, stream, input1, input2, output,
                     out_batch_stride, out_ch_stride, out_px_stride, ic, ih, iw, oc, oh, ow, pad_size, kernel_size, max_displacement, stride1, stride2);
  DTT_CHECK_LAUNCH("corr_fwd_generic");
  return 1;
}

extern "C" int dtt_correlation_backward(const float* gradOutput, int gob, int goc, int goh, int gow,
                                        const float* input1, int ic, int ih, int iw, const float* input2,
                                        float* gradInput1, float* gradInput2, int pad_size, int kernel_size,
                                        int max_displacement, int stride1, int stride2, int corr_type_multiply,
                                        void* stream_) {
  (void)corr_type_multiply;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gradOutput && input1 && input2 && gradInput1 && gradInput2, "correlation backward: null pointer");
  int eoc, eoh, eow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &eoc, &eoh, &eow))
    return 0;
  DTT_REQUIRE(gob > 0 && goc == eoc && goh == eoh && gow == eow, "correlation backward: gradOutput shape mismatch");
  int nbr;
  if (fast_path(kernel_size, stride1, stride2, max_displacement, &nbr) && nbr <= 5 && ic % 16 == 0) {
    const FastGeom g = make_geom(gob, ic, ih, iw, goc, goh, gow, pad_size, max_displacement, stride1);
    return nbr == 3 ? launch_bwd<3, 2, false>(gradOutput, input1, input2, gradInput1, gradInput2, g, gob, stream)
                    : launch_bwd<5, 2, false>(gradOutput, input1, input2, gradInput1, gradInput2, g, gob, stream);
  }
  const long total = (long)gob * ic * ih * iw;
  hipLaunchKernelGGL(corr_bwd_simple, dim3(min(dtt_cdiv(total, 256), 1 << 20)), dim3(256), 0, stream, gradOutput,
                     input1, input2, gradInput1, gradInput2, gob, ic, ih, iw, goc, goh, gow, pad_size, kernel_size,
                     max_displacement, stride1, stride2);
  DTT_CHECK_LAUNCH("corr_bwd_simple");
  return 1;
}

// Band-stationary, halo-streamed gradient kernels for channels-last maps (correlation_bwd.hip)
int dtt_corr_bwd_stream(const float* gradOutput, long g_sb, long g_sc, long g_sp, int gob, int goh, int gow, const float* input1, int ic,
                        int ih, int iw, const float* input2, float* gradInput1, float* gradInput2, int pad_size, int max_displacement,
                        int stride, int which, int phase, void* workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" int dtt_correlation_backward_stream_supported(int ic, int kernel_size, int max_displacement, int stride1, int stride2);

static int backward_nhwc_checks(const float* gradOutput, int gob, int goc, int goh, int gow, const float* input1, int ic, int ih, int iw,
                                const float* input2, const float* gradInput1, const float* gradInput2, int pad_size, int kernel_size,
                                int max_displacement, int stride1, int stride2, int* nbr, int max_nbr) {
  DTT_REQUIRE(gradOutput && input1 && input2, "correlation backward: null pointer");
  int eoc, eoh, eow;
  if (!dtt_correlation_output_shape(ic, ih, iw, pad_size, kernel_size, max_displacement, stride1, stride2, &eoc, &eoh, &eow))
    return 0;
  DTT_REQUIRE(gob > 0 && goc == eoc && goh == eoh && gow == eow, "correlation backward: gradOutput shape mismatch");
  DTT_REQUIRE(fast_path(kernel_size, stride1, stride2, max_displacement, nbr) && *nbr <= max_nbr && ic % 16 == 0,
              "correlation backward (channels-last): needs kernel_size 1, stride1 == stride2, max_displacement / stride <= %d and "
              "channels %% 16 == 0 (got k=%d s1=%d s2=%d d=%d C=%d)", max_nbr <= 5 ? 8 : 16, kernel_size, stride1, stride2, max_displacement, ic);
  DTT_REQUIRE(((reinterpret_cast<uintptr_t>(input1) | reinterpret_cast<uintptr_t>(input2) | reinterpret_cast<uintptr_t>(gradInput1) |
                reinterpret_cast<uintptr_t>(gradInput2)) & 15) == 0, "correlation backward (channels-last): pointers must be 16-byte aligned");
  return 1;
}

// Both gradients for channels-last inputs (n, ih, iw, ic) and channels-last gradInputs on the band-stationary streamed kernels
// (correlation_bwd.hip; kernel_size 1, stride1 == stride2, max_displacement / stride <= 16, ic % 64 == 0 -- conv3 / conv4 / conv5 of
// D&T at d = 8 and at BASELINE configs[4]'s d = 16; dtt_correlation_backward_stream_supported).  gradOut[n, d, p] is read at gradOutput[n * g_batch_stride + d * g_ch_stride +
// p * g_px_stride] (p = oy * ow + ox): the reference's (n, D*D, oh, ow) planes (strides D*D*oh*ow, oh*ow, 1) or columns of
// position-major rows (g_ch_stride = 1, g_px_stride = the row length).  which: 1 = gradInput1 only, 2 = gradInput2 only, 3 = both.
// workspace: dtt_correlation_backward_workspace_bytes(...) bytes, caller-owned, overwritten.
// phase: 1 = lay out the band words in the workspace only (reads gradOutput; event tag corr_bwd_band), 2 = the gradients from a
// workspace that a phase-1 call with the same arguments filled (event tag corr_bwd_op), 3 = both.  The two phases may run on
// different streams (the caller orders them): the band launches of several correlations then run beside another op.
extern "C" int dtt_correlation_backward_nhwc_phase(const float* gradOutput, long g_batch_stride, long g_ch_stride, long g_px_stride,
                                                   int gob, int goc, int goh, int gow, const float* input1, int ic, int ih, int iw,
                                                   const float* input2, float* gradInput1, float* gradInput2, int pad_size,
                                                   int kernel_size, int max_displacement, int stride1, int stride2, int which, int phase,
                                                   void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(phase >= 1 && phase <= 3, "correlation backward: phase %d (1 band words, 2 gradients, 3 both)", phase);
  DTT_REQUIRE(which >= 1 && which <= 3 && (phase == 1 || (((which & 1) == 0 || gradInput1) && ((which & 2) == 0 || gradInput2))),
              "correlation backward: null gradient pointer / bad selector");      // (phase 1 writes no gradient)
  int nbr = 0;
  if (!backward_nhwc_checks(gradOutput, gob, goc, goh, gow, input1, ic, ih, iw, input2, gradInput1, gradInput2, pad_size, kernel_size,
                            max_displacement, stride1, stride2, &nbr, 9))
    return 0;
  DTT_REQUIRE(dtt_correlation_backward_stream_supported(ic, kernel_size, max_displacement, stride1, stride2),
              "correlation backward (channels-last, streamed): needs channels %% 64 == 0 (got %d); use dtt_correlation_backward_nhwc", ic);
  const char* tag = phase == 1 ? "corr_bwd_band" : "corr_bwd_op";
  dtt_prof_begin(tag, stream);
  const int ok = dtt_corr_bwd_stream(gradOutput, g_batch_stride, g_ch_stride, g_px_stride, gob, goh, gow, input1, ic, ih, iw, input2,
                                     gradInput1, gradInput2, pad_size, max_displacement, stride1, which, phase, workspace, workspace_bytes,
                                     stream);
  dtt_prof_end(tag, stream);
  return ok;
}

extern "C" int dtt_correlation_backward_nhwc_strided(const float* gradOutput, long g_batch_stride, long g_ch_stride, long g_px_stride,
                                                     int gob, int goc, int goh, int gow, const float* input1, int ic, int ih, int iw,
                                                     const float* input2, float* gradInput1, float* gradInput2, int pad_size,
                                                     int kernel_size, int max_displacement, int stride1, int stride2, int which,
                                                     void* workspace, size_t workspace_bytes, void* stream_) {
  return dtt_correlation_backward_nhwc_phase(gradOutput, g_batch_stride, g_ch_stride, g_px_stride, gob, goc, goh, gow, input1, ic, ih, iw,
                                             input2, gradInput1, gradInput2, pad_size, kernel_size, max_displacement, stride1, stride2, which,
                                             3, workspace, workspace_bytes, stream_);
}

// Round 1's channels-last gradient kernels (corr_bwd_mfma<.., NHWC = true>): gradOutput the reference's contiguous (n, D*D, oh, ow),
// both gradients, no workspace; ic % 16 == 0.  The fallback for channel counts the streamed kernels do not take, and their A/B.
extern "C" int dtt_correlation_backward_nhwc(const float* gradOutput, int gob, int goc, int goh, int gow,
                                             const float* input1, int ic, int ih, int iw, const float* input2,
                                             float* gradInput1, float* gradInput2, int pad_size, int kernel_size,
                                             int max_displacement, int stride1, int stride2, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gradInput1 && gradInput2, "correlation backward: null pointer");
  int nbr = 0;
  if (!backward_nhwc_checks(gradOutput, gob, goc, goh, gow, input1, ic, ih, iw, input2, gradInput1, gradInput2, pad_size, kernel_size,
                            max_displacement, stride1, stride2, &nbr, 5))
    return 0;
  const FastGeom g = make_geom(gob, ic, ih, iw, goc, goh, gow, pad_size, max_displacement, stride1);
  dtt_prof_begin("corr_bwd_op", stream);
  const int ok = nbr == 3 ? launch_bwd<3, 2, true>(gradOutput, input1, input2, gradInput1, gradInput2, g, gob, stream)
                          : launch_bwd<5, 2, true>(gradOutput, input1, input2, gradInput1, gradInput2, g, gob, stream);
  dtt_prof_end("corr_bwd_op", stream);
  return ok;
}

// Weight gradient of the 1x1 heads for gfx950: dW[n, k] = sum_m G[m, n] * X[m, k] over the pixel rows m of the position-major
// maps (rfcn.py:49-53 in the training graph: RFCN_cls_net + RFCN_bbox_net packed as one 1776 x 512 filter, 10 184 pixel rows
// for two frame pairs at 600 px).
//
// Both operands are already "m-major": an exact-f32 v_mfma_f32_16x16x4_f32 wants A = 16 outputs x 4 pixels and B = 4 pixels x
// 16 inputs, i.e. lane l reads G[m + l / 16][n0 + l % 16] and X[m + l / 16][k0 + l % 16] -- 64-byte runs of the rows as they
// lie in memory.  No operand is transposed (round 3's first version transposed both with the tiled transpose and ran the
// forward kernel over them: 436 + 32 us, half of its waves idle on the short 1776 x 512 output).
//   * workgroup = 8 waves, a 256 (outputs) x 128 (inputs) tile of dW, one slice of the pixel rows; wave (wy, wx) owns a
//     64 x 64 sub-tile = 4 x 4 MFMA tiles, 64 accumulator registers, the whole slice accumulated in registers;
//   * per chunk of 32 pixel rows the tile's 32 x 256 piece of G and 32 x 128 piece of X go global -> registers -> LDS (double
//     buffered: the loads of chunk i + 1 are in flight under the 128 MFMAs per wave of chunk i; one barrier per chunk).  LDS rows
//     are padded by 16 floats so that the four pixel rows of a fragment read fall on the two halves of the banks: 2 cycles
//     per ds_read_b32, the minimum for 64 lanes;
//   * the pixel rows are split over `slices` workgroups per tile so that tiles x slices fills the chip in ONE round (7 x 4 tiles x
//     9 slices = 252 workgroups at the D&T shape; ten slices would be 280: a second, nearly empty round as long as the first --
//     283 us against 164); the slices' partial tiles go to a workspace and a second kernel adds them IN SLICE ORDER:
//     deterministic, no atomics, no tickets.
// Measured (training step at 600 px, rocprofv3): 164 + 6 us = 113 TFLOP/s, 0.72 of the fp32 MFMA peak.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBN = 256, kBK = 128, kBM = 32;       // tile of dW (outputs x inputs), pixel rows per chunk
constexpr int kWaves = 8, kThreads = kWaves * 64;
constexpr int kLdG = kBN + 16, kLdX = kBK + 16;     // LDS row strides (floats): = 16 mod 32
constexpr int kBufFloats = kBM * (kLdG + kLdX);     // one stage
constexpr int kGVec = kBM * kBN / 4 / kThreads;     // float4 loads per thread per chunk: G (4), X (2)
constexpr int kXVec = kBM * kBK / 4 / kThreads;

struct DwGeom {
  const float* g; long ldg;     // gradient rows (M, >= n_cols), row stride in floats
  const float* x; long ldx;     // input rows (M, K)
  int M, n_cols, N, K;          // pixel rows; readable columns of g (its row stride's worth); outputs stored; inputs
  int tiles_n, tiles_k, slices, chunks;   // chunks = ceil(M / 32), dealt to the slices in contiguous runs
  float* ws;                    // [slices][tiles_n * 256][K]
};

__global__ __launch_bounds__(kThreads) void head_dw_kernel(DwGeom d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int item = dtt_xcd_remap(blockIdx.x, gridDim.x);
  const int tile = item % (d.tiles_n * d.tiles_k), sl = item / (d.tiles_n * d.tiles_k);
  const int tn = tile / d.tiles_k, tk = tile - tn * d.tiles_k;
  const int n0 = tn * kBN, k0 = tk * kBK;
  const int c_lo = (int)((long)d.chunks * sl / d.slices), c_hi = (int)((long)d.chunks * (sl + 1) / d.slices);
  const int wy = wave >> 1, wx = wave & 1;            // 4 x 2 waves: rows of 64 outputs, columns of 64 inputs

  // staging slots of this thread: G piece (row r, float4 column q of 64), X piece (row r, float4 column q of 32)
  int g_row[kGVec], g_col[kGVec], x_row[kXVec], x_col[kXVec];
#pragma unroll
  for (int i = 0; i < kGVec; ++i) { const int f = tid + i * kThreads; g_row[i] = f >> 6; g_col[i] = (f & 63) * 4; }
#pragma unroll
  for (int i = 0; i < kXVec; ++i) { const int f = tid + i * kThreads; x_row[i] = f >> 5; x_col[i] = (f & 31) * 4; }
  f32x4 gv[kGVec], xv[kXVec];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int chunk) {
    const int m0 = chunk * kBM;
#pragma unroll
    for (int i = 0; i < kGVec; ++i) {
      const int m = m0 + g_row[i], n = n0 + g_col[i];
      const bool ok = m < d.M && n + 3 < d.n_cols;      // (n_cols % 4 == 0: a float4 is wholly inside or outside)
      gv[i] = ok ? *reinterpret_cast<const f32x4*>(d.g + (long)m * d.ldg + n) : zero4;
    }
#pragma unroll
    for (int i = 0; i < kXVec; ++i) {
      const int m = m0 + x_row[i], k = k0 + x_col[i];
      const bool ok = m < d.M && k + 3 < d.K;
      xv[i] = ok ? *reinterpret_cast<const f32x4*>(d.x + (long)m * d.ldx + k) : zero4;
    }
  };
  auto store = [&](int buf) {
    float* G = lds + buf * kBufFloats;
    float* X = G + kBM * kLdG;
#pragma unroll
    for (int i = 0; i < kGVec; ++i) *reinterpret_cast<f32x4*>(G + g_row[i] * kLdG + g_col[i]) = gv[i];
#pragma unroll
    for (int i = 0; i < kXVec; ++i) *reinterpret_cast<f32x4*>(X + x_row[i] * kLdX + x_col[i]) = xv[i];
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = zero4;

  if (c_lo < c_hi) {
    issue(c_lo);
    store(0);
    __syncthreads();
    const int a_off = (lane >> 4) * kLdG + wy * 64 + (lane & 15);   // A[i = output (lane % 16)][kk = pixel row (lane / 16)]
    const int b_off = (lane >> 4) * kLdX + wx * 64 + (lane & 15);   // B[kk = pixel row][j = input (lane % 16)]
    for (int c = c_lo; c < c_hi; ++c) {
      const int buf = (c - c_lo) & 1;
      const bool more = c + 1 < c_hi;
      if (more) issue(c + 1);
      const float* G = lds + buf * kBufFloats;
      const float* X = G + kBM * kLdG;
#pragma unroll
      for (int ms = 0; ms < kBM / 4; ++ms) {
        float af[4], bf[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) af[a] = G[a_off + ms * 4 * kLdG + a * 16];
#pragma unroll
        for (int b = 0; b < 4; ++b) bf[b] = X[b_off + ms * 4 * kLdX + b * 16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
      }
      if (more) store(buf ^ 1);
      __syncthreads();
    }
  }
  // D[i = 4 * (lane / 16) + r][j = lane % 16] of MFMA tile (a, b) -> ws[sl][n0 + wy * 64 + a * 16 + i][k0 + wx * 64 + b * 16 + j]
  float* out = d.ws + ((long)sl * d.tiles_n * kBN + n0 + wy * 64) * d.K + k0 + wx * 64;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = a * 16 + (lane >> 4) * 4 + r;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = k0 + wx * 64 + b * 16 + (lane & 15);
        if (k < d.K) out[(long)i * d.K + b * 16 + (lane & 15)] = acc[a][b][r];
      }
    }
}

// dw[n][k] = sum over the slices, in slice order; one float4 per thread
__global__ __launch_bounds__(256) void head_dw_reduce(const float* __restrict__ ws, long slice_floats, int slices, float* __restrict__ dw,
                                                      long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  f32x4 s = reinterpret_cast<const f32x4*>(ws)[i];
  for (int k = 1; k < slices; ++k) {
    const f32x4 v = reinterpret_cast<const f32x4*>(ws + k * slice_floats)[i];
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  reinterpret_cast<f32x4*>(dw)[i] = s;
}

int dw_slices(int M, int N, int K) {
  const int tiles = ((N + kBN - 1) / kBN) * ((K + kBK - 1) / kBK), chunks = (M + kBM - 1) / kBM;
  int s = dtt_device_cus() / tiles;                   // ONE round of workgroups over the chip
  if (s > chunks) s = chunks;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t dtt_head_gemm_dw_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)dw_slices(M, N, K) * ((N + kBN - 1) / kBN) * kBN * (size_t)K * sizeof(float);
}

extern "C" int dtt_head_gemm_dw(const float* gout, long ldg, int g_cols, const float* x, long ldx, int M, int N, int K, float* dw,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(gout && x && dw, "head_gemm_dw: null pointer");
  DTT_REQUIRE(M > 0 && N > 0 && K > 0 && g_cols >= N && ldg >= g_cols && ldx >= K, "head_gemm_dw: bad shape");
  DTT_REQUIRE(K % 4 == 0 && g_cols % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0, "head_gemm_dw: K, the gradient's columns and both row strides must be multiples of 4");
  DTT_REQUIRE(((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dw) |
                reinterpret_cast<uintptr_t>(workspace)) & 15) == 0, "head_gemm_dw: pointers must be 16-byte aligned");
  DwGeom d;
  d.g = gout; d.ldg = ldg; d.x = x; d.ldx = ldx; d.M = M; d.n_cols = g_cols; d.N = N; d.K = K;
  d.tiles_n = (N + kBN - 1) / kBN; d.tiles_k = (K + kBK - 1) / kBK;
  d.slices = dw_slices(M, N, K); d.chunks = (M + kBM - 1) / kBM;
  const size_t need = dtt_head_gemm_dw_workspace_bytes(M, N, K);
  DTT_REQUIRE(workspace && workspace_bytes >= need, "head_gemm_dw: workspace too small (%zu < %zu)", workspace_bytes, need);
  d.ws = static_cast<float*>(workspace);
  const size_t lds = (size_t)2 * kBufFloats * sizeof(float);
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_dw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DTT_REQUIRE(e == hipSuccess, "head_gemm_dw: cannot raise dynamic LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL(head_dw_kernel, dim3(d.tiles_n * d.tiles_k * d.slices), dim3(kThreads), lds, stream, d);
  DTT_CHECK_LAUNCH("head_dw_kernel");
  // the first N rows of the padded tile rows are the result (rows N .. tiles_n * 256 of a slice are products with padding columns)
  const long slice_floats = (long)d.tiles_n * kBN * K, total4 = (long)N * K / 4;
  hipLaunchKernelGGL(head_dw_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, d.ws, slice_floats, d.slices, dw, total4);
  DTT_CHECK_LAUNCH("head_dw_reduce");
  return 1;
}

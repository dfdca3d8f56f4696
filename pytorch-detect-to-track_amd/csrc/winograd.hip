// Winograd F(2x2, 3x3) / F(4x4, 3x3) transforms for the channels-last trunk's 3x3, stride-1 convolutions (gfx950).
//
// The frozen-BatchNorm ResNet trunk spends half of the inference step in fp32 3x3 convolutions that MIOpen's implicit
// GEMMs already run at 70-89 % of the fp32 MFMA peak; the lever left inside fp32 arithmetic is fewer multiplies.
// out = A^T [ sum_c (G g G^T) .* (B^T d B) ] A turns a 3x3 convolution into (m + 2)^2 independent GEMMs over m x m output
// tiles: 16 GEMMs and 2.25x fewer MACs for m = 2, 36 GEMMs and 4x fewer MACs for m = 4 (interpolation points 0, +-1, +-2,
// infinity; the variant cuDNN's fp32 WINOGRAD_NONFUSED uses).  The GEMMs are library calls (dtt_gemm_batched, hipBLASLt, candidates timed per shape); this
// file holds the two memory-bound transforms around them:
//   input :  x (N,H,W,C) channels-last  ->  V[(m+2)^2][tiles][C]  (zero padding folded in)
//   output:  M[(m+2)^2][tiles][K]       ->  y (N,H,W,K) channels-last, + bias[k], optional ReLU (the trunk's epilogue)
// A dilated convolution (dilation d, padding d) is d*d interleaved ordinary convolutions on the parity sub-lattices, so a
// tile is (image, py, px, ty, tx) and all of them go through the same GEMMs.
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct WinoGeom {
  int N, H, W, d;      // image count, height, width, dilation
  int m;               // output tile edge: 2 = F(2x2, 3x3), 4 = F(4x4, 3x3)
  int ths, tws;        // tiles per sub-lattice (rows, cols), sized for the largest sub-lattice
  long tiles;          // N * d * d * ths * tws
};

__device__ __forceinline__ void decode_tile(const WinoGeom& g, long t, int& n, int& py, int& px, int& ty, int& tx) {
  tx = (int)(t % g.tws); t /= g.tws;
  ty = (int)(t % g.ths); t /= g.ths;
  px = (int)(t % g.d); t /= g.d;
  py = (int)(t % g.d); t /= g.d;
  n = (int)t;
}

// VW consecutive channels per thread (4: one 16-byte access; 1 / 2: more threads and fewer registers per thread, for the
// launches that would otherwise not fill the chip -- a F(4x4, 3x3) thread holds 36 values per channel)
template <int VW>
struct vec {
  float v[VW];
};
template <int VW>
__device__ __forceinline__ vec<VW> operator+(vec<VW> a, vec<VW> b) {
  vec<VW> r;
#pragma unroll
  for (int i = 0; i < VW; ++i) r.v[i] = a.v[i] + b.v[i];
  return r;
}
template <int VW>
__device__ __forceinline__ vec<VW> operator-(vec<VW> a, vec<VW> b) {
  vec<VW> r;
#pragma unroll
  for (int i = 0; i < VW; ++i) r.v[i] = a.v[i] - b.v[i];
  return r;
}
template <int VW>
__device__ __forceinline__ vec<VW> operator*(float s, vec<VW> a) {
  vec<VW> r;
#pragma unroll
  for (int i = 0; i < VW; ++i) r.v[i] = s * a.v[i];
  return r;
}
template <int VW>
__device__ __forceinline__ vec<VW> ldv(const float* p) {
  vec<VW> r;
  __builtin_memcpy(&r, __builtin_assume_aligned(p, 4 * VW), 4 * VW);
  return r;
}
template <int VW>
__device__ __forceinline__ void stv(float* p, vec<VW> v) { __builtin_memcpy(__builtin_assume_aligned(p, 4 * VW), &v, 4 * VW); }
template <int VW>
__device__ __forceinline__ vec<VW> zerov() {
  vec<VW> r;
#pragma unroll
  for (int i = 0; i < VW; ++i) r.v[i] = 0.f;
  return r;
}

// B^T applied to one line of M + 2 values, in place
template <int M, typename V>
__device__ __forceinline__ void bt_line(V* v, int s) {
  if constexpr (M == 2) {   // rows (1 0 -1 0), (0 1 1 0), (0 -1 1 0), (0 1 0 -1)
    const V d0 = v[0], d1 = v[s], d2 = v[2 * s], d3 = v[3 * s];
    v[0] = d0 - d2; v[s] = d1 + d2; v[2 * s] = d2 - d1; v[3 * s] = d1 - d3;
  } else {   // (4 0 -5 0 1 0), (0 -4 -4 1 1 0), (0 4 -4 -1 1 0), (0 -2 -1 2 1 0), (0 2 -1 -2 1 0), (0 4 0 -5 0 1)
    const V d0 = v[0], d1 = v[s], d2 = v[2 * s], d3 = v[3 * s], d4 = v[4 * s], d5 = v[5 * s];
    const V a = d4 - 4.f * d2, b = d3 - 4.f * d1, c = d4 - d2, e = 2.f * (d3 - d1);
    v[0] = 4.f * d0 - 5.f * d2 + d4;
    v[s] = a + b; v[2 * s] = a - b;
    v[3 * s] = c + e; v[4 * s] = c - e;
    v[5 * s] = 4.f * d1 - 5.f * d3 + d5;
  }
}

// A^T applied to one line of M + 2 values -> M values (written to out[0 .. M-1] with stride so)
template <int M, typename V>
__device__ __forceinline__ void at_line(const V* v, int s, V* out, int so) {
  if constexpr (M == 2) {   // (1 1 1 0), (0 1 -1 -1)
    out[0] = v[0] + v[s] + v[2 * s];
    out[so] = v[s] - v[2 * s] - v[3 * s];
  } else {   // (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1)
    const V p = v[s] + v[2 * s], q = v[s] - v[2 * s], r = v[3 * s] + v[4 * s], t = v[3 * s] - v[4 * s];
    out[0] = v[0] + p + r;
    out[so] = q + 2.f * t;
    out[2 * so] = p + 4.f * r;
    out[3 * so] = q + 8.f * t + v[5 * s];
  }
}

// one thread = one (tile, VW channels): (M+2)^2 guarded loads, B^T d B, (M+2)^2 stores
template <int M, int VW>
__global__ __launch_bounds__(kThreads) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, WinoGeom g,
                                                              int C) {
  constexpr int T = M + 2;
  using vt = vec<VW>;
  const int cn = C / VW;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= g.tiles * cn) return;
  const int c = (int)(idx % cn) * VW;
  const long tile = idx / cn;
  int n, py, px, ty, tx;
  decode_tile(g, tile, n, py, px, ty, tx);
  vt d[T * T];
#pragma unroll
  for (int a = 0; a < T; ++a) {
    const int iy = (M * ty - 1 + a) * g.d + py;
#pragma unroll
    for (int b = 0; b < T; ++b) {
      const int ix = (M * tx - 1 + b) * g.d + px;
      const bool in = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
      d[a * T + b] = in ? ldv<VW>(x + (((long)n * g.H + iy) * g.W + ix) * C + c) : zerov<VW>();
    }
  }
#pragma unroll
  for (int b = 0; b < T; ++b) bt_line<M>(d + b, T);        // columns: B^T d
#pragma unroll
  for (int a = 0; a < T; ++a) bt_line<M>(d + a * T, 1);    // rows: (B^T d) B
  const long stride = g.tiles * C;
  float* dst = V + tile * C + c;
#pragma unroll
  for (int i = 0; i < T * T; ++i) stv<VW>(dst + i * stride, d[i]);
}

// one thread = one (tile, VW output channels): (M+2)^2 loads, A^T m A, bias (+ ReLU), up to M^2 guarded stores
template <int M, int VW, bool RELU>
__global__ __launch_bounds__(kThreads) void wino_output_kernel(const float* __restrict__ Mm, const float* __restrict__ bias,
                                                               float* __restrict__ y, WinoGeom g, int K) {
  constexpr int T = M + 2;
  using vt = vec<VW>;
  const int kn = K / VW;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= g.tiles * kn) return;
  const int k = (int)(idx % kn) * VW;
  const long tile = idx / kn;
  int n, py, px, ty, tx;
  decode_tile(g, tile, n, py, px, ty, tx);
  const long stride = g.tiles * K;
  const float* src = Mm + tile * K + k;
  vt m[T * T];
#pragma unroll
  for (int i = 0; i < T * T; ++i) m[i] = ldv<VW>(src + i * stride);
  vt t[M * T];   // A^T m : M rows x T columns
#pragma unroll
  for (int b = 0; b < T; ++b) at_line<M>(m + b, T, t + b, T);
  const vt bv = ldv<VW>(bias + k);
#pragma unroll
  for (int i = 0; i < M; ++i) {
    vt o[M];
    at_line<M>(t + i * T, 1, o, 1);
    const int oy = (M * ty + i) * g.d + py;
    if (oy >= g.H) continue;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const int ox = (M * tx + j) * g.d + px;
      if (ox >= g.W) continue;
      vt v = o[j] + bv;
      if (RELU) {
#pragma unroll
        for (int q = 0; q < VW; ++q) v.v[q] = fmaxf(v.v[q], 0.f);
      }
      stv<VW>(y + (((long)n * g.H + oy) * g.W + ox) * K + k, v);
    }
  }
}

int make_geom(int images, int height, int width, int dilation, int m, WinoGeom& g) {
  DTT_REQUIRE(images > 0 && height > 0 && width > 0 && dilation > 0, "winograd: bad shape");
  DTT_REQUIRE(m == 2 || m == 4, "winograd: output tile must be 2 or 4 (got %d)", m);
  g.N = images; g.H = height; g.W = width; g.d = dilation; g.m = m;
  const int hs = (height + dilation - 1) / dilation, ws = (width + dilation - 1) / dilation;
  g.ths = (hs + m - 1) / m; g.tws = (ws + m - 1) / m;
  g.tiles = (long)images * dilation * dilation * g.ths * g.tws;
  return 1;
}

}  // namespace

extern "C" long dtt_winograd_tiles(int images, int height, int width, int dilation, int m) {
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, m, g)) return -1;
  return g.tiles;
}

extern "C" int dtt_winograd_input_transform(const float* x, float* v, int images, int height, int width, int channels,
                                            int dilation, int m, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(x && v, "winograd input transform: null pointer");
  DTT_REQUIRE(channels > 0 && channels % 4 == 0, "winograd: channels must be a multiple of 4 (got %d)", channels);
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, m, g)) return 0;
  // 16-byte accesses when that still gives every CU a few workgroups, otherwise one channel per thread
  const int vw = g.tiles * (channels / 4) >= 4L * 256 * kThreads ? 4 : 1;
  const dim3 grid((unsigned)dtt_cdiv(g.tiles * (channels / vw), kThreads));
  if (m == 2 && vw == 4) hipLaunchKernelGGL((wino_input_kernel<2, 4>), grid, dim3(kThreads), 0, stream, x, v, g, channels);
  else if (m == 2) hipLaunchKernelGGL((wino_input_kernel<2, 1>), grid, dim3(kThreads), 0, stream, x, v, g, channels);
  else if (vw == 4) hipLaunchKernelGGL((wino_input_kernel<4, 4>), grid, dim3(kThreads), 0, stream, x, v, g, channels);
  else hipLaunchKernelGGL((wino_input_kernel<4, 1>), grid, dim3(kThreads), 0, stream, x, v, g, channels);
  DTT_CHECK_LAUNCH("wino_input_kernel");
  return 1;
}

extern "C" int dtt_winograd_output_transform(const float* mm, const float* bias, float* y, int images, int height, int width,
                                             int channels, int dilation, int m, int relu, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(mm && bias && y, "winograd output transform: null pointer");
  DTT_REQUIRE(channels > 0 && channels % 4 == 0, "winograd: channels must be a multiple of 4 (got %d)", channels);
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, m, g)) return 0;
  const int vw = g.tiles * (channels / 4) >= 4L * 256 * kThreads ? 4 : 1;
  const dim3 grid((unsigned)dtt_cdiv(g.tiles * (channels / vw), kThreads));
#define DTT_WINO_OUT(MM, VW, RL) \
  hipLaunchKernelGGL((wino_output_kernel<MM, VW, RL>), grid, dim3(kThreads), 0, stream, mm, bias, y, g, channels)
  if (m == 2 && vw == 4) { if (relu) DTT_WINO_OUT(2, 4, true); else DTT_WINO_OUT(2, 4, false); }
  else if (m == 2) { if (relu) DTT_WINO_OUT(2, 1, true); else DTT_WINO_OUT(2, 1, false); }
  else if (vw == 4) { if (relu) DTT_WINO_OUT(4, 4, true); else DTT_WINO_OUT(4, 4, false); }
  else { if (relu) DTT_WINO_OUT(4, 1, true); else DTT_WINO_OUT(4, 1, false); }
#undef DTT_WINO_OUT
  DTT_CHECK_LAUNCH("wino_output_kernel");
  return 1;
}

// Winograd F(2x2, 3x3) transforms for the channels-last trunk's 3x3, stride-1 convolutions (gfx950).
//
// The frozen-BatchNorm ResNet trunk spends half of the inference step in fp32 3x3 convolutions that MIOpen's implicit
// GEMMs already run at 70-89 % of the fp32 MFMA peak; the lever left inside fp32 arithmetic is fewer multiplies.
// out = A^T [ sum_c (G g G^T) .* (B^T d B) ] A turns a 3x3 convolution into 16 independent GEMMs over 2x2-output tiles
// with 2.25x fewer MACs.  The GEMMs are library calls (dtt_gemm_batched, hipBLASLt, candidates timed per shape); this
// file holds the two memory-bound transforms around them:
//   input :  x (N,H,W,C) channels-last  ->  V[16][tiles][C]       (zero padding folded in)
//   output:  M[16][tiles][K]            ->  y (N,H,W,K) channels-last, + bias[k], optional ReLU (the trunk's epilogue)
// A dilated convolution (dilation d, padding d) is d*d interleaved ordinary convolutions on the parity sub-lattices, so a
// tile is (image, py, px, ty, tx) and all of them go through the same 16 GEMMs.
#include "common.h"

namespace {

constexpr int kThreads = 256;

struct WinoGeom {
  int N, H, W, d;      // image count, height, width, dilation
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

// one thread = one (tile, 4 channels): 16 guarded float4 loads, B^T d B, 16 float4 stores
__global__ __launch_bounds__(kThreads) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, WinoGeom g,
                                                              int C) {
  const int c4n = C >> 2;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= g.tiles * c4n) return;
  const int c = (int)(idx % c4n) << 2;
  const long tile = idx / c4n;
  int n, py, px, ty, tx;
  decode_tile(g, tile, n, py, px, ty, tx);
  float4 dmat[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int iy = (2 * ty - 1 + a) * g.d + py;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ix = (2 * tx - 1 + b) * g.d + px;
      const bool in = iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
      dmat[a][b] = in ? *reinterpret_cast<const float4*>(x + (((long)n * g.H + iy) * g.W + ix) * C + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto sub = [](float4 p, float4 q) { return make_float4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w); };
  auto add = [](float4 p, float4 q) { return make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w); };
  float4 tmp[4][4];   // B^T d : rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    tmp[0][b] = sub(dmat[0][b], dmat[2][b]);
    tmp[1][b] = add(dmat[1][b], dmat[2][b]);
    tmp[2][b] = sub(dmat[2][b], dmat[1][b]);
    tmp[3][b] = sub(dmat[1][b], dmat[3][b]);
  }
  const long stride = g.tiles * C;
  float* dst = V + tile * C + c;
#pragma unroll
  for (int a = 0; a < 4; ++a) {   // (B^T d) B : same combination along columns
    *reinterpret_cast<float4*>(dst + (a * 4 + 0) * stride) = sub(tmp[a][0], tmp[a][2]);
    *reinterpret_cast<float4*>(dst + (a * 4 + 1) * stride) = add(tmp[a][1], tmp[a][2]);
    *reinterpret_cast<float4*>(dst + (a * 4 + 2) * stride) = sub(tmp[a][2], tmp[a][1]);
    *reinterpret_cast<float4*>(dst + (a * 4 + 3) * stride) = sub(tmp[a][1], tmp[a][3]);
  }
}

// one thread = one (tile, 4 output channels): 16 float4 loads, A^T m A, bias (+ ReLU), up to 4 guarded float4 stores
template <bool RELU>
__global__ __launch_bounds__(kThreads) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                               float* __restrict__ y, WinoGeom g, int K) {
  const int k4n = K >> 2;
  const long idx = (long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= g.tiles * k4n) return;
  const int k = (int)(idx % k4n) << 2;
  const long tile = idx / k4n;
  int n, py, px, ty, tx;
  decode_tile(g, tile, n, py, px, ty, tx);
  const long stride = g.tiles * K;
  const float* src = M + tile * K + k;
  float4 m[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) m[a][b] = *reinterpret_cast<const float4*>(src + (a * 4 + b) * stride);
  auto add3 = [](float4 p, float4 q, float4 r) { return make_float4(p.x + q.x + r.x, p.y + q.y + r.y, p.z + q.z + r.z, p.w + q.w + r.w); };
  auto sub3 = [](float4 p, float4 q, float4 r) { return make_float4(p.x - q.x - r.x, p.y - q.y - r.y, p.z - q.z - r.z, p.w - q.w - r.w); };
  float4 t0[4], t1[4];   // A^T m : rows (m0 + m1 + m2, m1 - m2 - m3)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t0[b] = add3(m[0][b], m[1][b], m[2][b]);
    t1[b] = sub3(m[1][b], m[2][b], m[3][b]);
  }
  const float4 bv = *reinterpret_cast<const float4*>(bias + k);
  float4 o[2][2];
  o[0][0] = add3(t0[0], t0[1], t0[2]); o[0][1] = sub3(t0[1], t0[2], t0[3]);
  o[1][0] = add3(t1[0], t1[1], t1[2]); o[1][1] = sub3(t1[1], t1[2], t1[3]);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oy = (2 * ty + i) * g.d + py;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ox = (2 * tx + j) * g.d + px;
      if (oy >= g.H || ox >= g.W) continue;
      float4 v = make_float4(o[i][j].x + bv.x, o[i][j].y + bv.y, o[i][j].z + bv.z, o[i][j].w + bv.w);
      if (RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      *reinterpret_cast<float4*>(y + (((long)n * g.H + oy) * g.W + ox) * K + k) = v;
    }
  }
}

int make_geom(int images, int height, int width, int dilation, WinoGeom& g) {
  DTT_REQUIRE(images > 0 && height > 0 && width > 0 && dilation > 0, "winograd: bad shape");
  g.N = images; g.H = height; g.W = width; g.d = dilation;
  const int hs = (height + dilation - 1) / dilation, ws = (width + dilation - 1) / dilation;
  g.ths = (hs + 1) / 2; g.tws = (ws + 1) / 2;
  g.tiles = (long)images * dilation * dilation * g.ths * g.tws;
  return 1;
}

}  // namespace

extern "C" long dtt_winograd_tiles(int images, int height, int width, int dilation) {
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, g)) return -1;
  return g.tiles;
}

extern "C" int dtt_winograd_input_transform(const float* x, float* v, int images, int height, int width, int channels,
                                            int dilation, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(x && v, "winograd input transform: null pointer");
  DTT_REQUIRE(channels > 0 && channels % 4 == 0, "winograd: channels must be a multiple of 4 (got %d)", channels);
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, g)) return 0;
  const long n = g.tiles * (channels / 4);
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)dtt_cdiv(n, kThreads)), dim3(kThreads), 0, stream, x, v, g, channels);
  DTT_CHECK_LAUNCH("wino_input_kernel");
  return 1;
}

extern "C" int dtt_winograd_output_transform(const float* m, const float* bias, float* y, int images, int height, int width,
                                             int channels, int dilation, int relu, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(m && bias && y, "winograd output transform: null pointer");
  DTT_REQUIRE(channels > 0 && channels % 4 == 0, "winograd: channels must be a multiple of 4 (got %d)", channels);
  WinoGeom g;
  if (!make_geom(images, height, width, dilation, g)) return 0;
  const long n = g.tiles * (channels / 4);
  if (relu)
    hipLaunchKernelGGL(wino_output_kernel<true>, dim3((unsigned)dtt_cdiv(n, kThreads)), dim3(kThreads), 0, stream, m, bias, y, g, channels);
  else
    hipLaunchKernelGGL(wino_output_kernel<false>, dim3((unsigned)dtt_cdiv(n, kThreads)), dim3(kThreads), 0, stream, m, bias, y, g, channels);
  DTT_CHECK_LAUNCH("wino_output_kernel");
  return 1;
}

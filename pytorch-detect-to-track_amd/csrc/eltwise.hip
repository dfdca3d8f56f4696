// Fused per-channel bias (+ residual) (+ ReLU) epilogue for the frozen-BatchNorm trunk, gfx950.
//
// The reference runs conv -> BatchNorm (frozen, eval mode: resnet.py:290-295, 325-330) -> ReLU and, at the end of a
// bottleneck, conv -> BatchNorm -> add residual -> ReLU as separate elementwise passes (resnet.py:88-107): 7
// full-tensor passes per bottleneck.  With the BatchNorm affine folded into the convolution weights (exact
// algebra, done once at load time in dtt/fuse.py) what remains is y = act(x + bias[c] (+ residual)): one pass,
// in place, HBM-bound.  One workgroup row per (image, channel) plane so the bias is a scalar; 16-byte accesses on
// the 16-byte-aligned middle of each plane, scalar head / tail.
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void bias_act_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                            const float* __restrict__ res, int channels, int hw) {
  const int plane = blockIdx.y;  // n * channels + c
  const float b = bias[plane % channels];
  float* px = x + (long)plane * hw;
  const float* pr = RES ? res + (long)plane * hw : nullptr;
  // elements before the first 16-byte boundary of this plane
  const int mis = (int)((reinterpret_cast<uintptr_t>(px) >> 2) & 3);
  const int head = min(hw, (4 - mis) & 3);
  const int nvec = (hw - head) >> 2;
  const int tail0 = head + (nvec << 2);
  const int t = blockIdx.x * kThreads + threadIdx.x, stride = gridDim.x * kThreads;
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) {
      float v = px[threadIdx.x] + b;
      if (RES) v += pr[threadIdx.x];
      px[threadIdx.x] = RELU ? fmaxf(v, 0.f) : v;
    }
    const int ti = tail0 + (int)threadIdx.x;
    if ((int)threadIdx.x < hw - tail0) {
      float v = px[ti] + b;
      if (RES) v += pr[ti];
      px[ti] = RELU ? fmaxf(v, 0.f) : v;
    }
  }
  float4* vx = reinterpret_cast<float4*>(px + head);
  const bool res_aligned = RES ? ((reinterpret_cast<uintptr_t>(pr + head) & 15) == 0) : true;
  for (int i = t; i < nvec; i += stride) {
    float4 v = vx[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (RES) {
      float4 r;
      if (res_aligned) r = reinterpret_cast<const float4*>(pr + head)[i];
      else __builtin_memcpy(&r, pr + head + 4 * i, 16);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    vx[i] = v;
  }
}

// Row-major (rows, channels) variant for the channels-last trunk: the bias index is the fastest dimension.
template <bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void bias_act_rows_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, long n4, int c4) {
  const long stride = (long)gridDim.x * kThreads;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride) {
    const float4 b = reinterpret_cast<const float4*>(bias)[i % c4];
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    if (RES) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

// Batched 2-D transpose in (batch, rows, cols) -> out (batch, cols, rows): the layout change between the
// channels-last trunk and the NCHW maps the D&T operators read (NHWC -> NCHW: rows = H*W, cols = C; the reverse for
// gradients).  64 x 64 tile through LDS (stride 65: both phases bank-conflict free), 256-byte row segments on both
// sides; 16-byte accesses on whichever side has a multiple-of-4 fastest extent.
constexpr int kTile = 64;

template <bool VR, bool VW>
__global__ __launch_bounds__(kThreads) void transpose_tiles(const float* __restrict__ in, float* __restrict__ out,
                                                            int rows, int cols) {
  __shared__ float tile[kTile][kTile + 1];
  const int c0 = blockIdx.x * kTile, r0 = blockIdx.y * kTile, tid = threadIdx.x;
  const float* src = in + (long)blockIdx.z * rows * cols;
  float* dst = out + (long)blockIdx.z * rows * cols;
  if (VR) {
#pragma unroll
    for (int it = 0; it < kTile * kTile / 4 / kThreads; ++it) {
      const int idx = it * kThreads + tid, r = idx >> 4, c = (idx & 15) << 2;
      if (r0 + r < rows && c0 + c < cols) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long)(r0 + r) * cols + c0 + c);
        tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < kTile * kTile / kThreads; ++it) {
      const int idx = it * kThreads + tid, r = idx >> 6, c = idx & 63;
      if (r0 + r < rows && c0 + c < cols) tile[r][c] = src[(long)(r0 + r) * cols + c0 + c];
    }
  }
  __syncthreads();
  if (VW) {
#pragma unroll
    for (int it = 0; it < kTile * kTile / 4 / kThreads; ++it) {
      const int idx = it * kThreads + tid, c = idx >> 4, r = (idx & 15) << 2;
      if (c0 + c < cols && r0 + r < rows)
        *reinterpret_cast<float4*>(dst + (long)(c0 + c) * rows + r0 + r) =
            make_float4(tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]);
    }
  } else {
#pragma unroll
    for (int it = 0; it < kTile * kTile / kThreads; ++it) {
      const int idx = it * kThreads + tid, c = idx >> 6, r = idx & 63;
      if (c0 + c < cols && r0 + r < rows) dst[(long)(c0 + c) * rows + r0 + r] = tile[r][c];
    }
  }
}

// dst[r][dst_col + k * ncols + c] = src[k * src_block_rows + r][src_col + c]: column blocks of a row-major matrix gathered
// side by side (float4 per lane, one row segment per run of lanes)
__global__ __launch_bounds__(kThreads) void gather_column_blocks_kernel(float* __restrict__ dst, long dst_ld, const float* __restrict__ src,
                                                                       long src_ld, long src_block_rows, int n_blocks, long rows, int c4) {
  const long total = rows * n_blocks * c4;
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const long r = i / ((long)n_blocks * c4);
    const int rem = (int)(i - r * n_blocks * c4), k = rem / c4, c = rem - k * c4;
    reinterpret_cast<float4*>(dst + r * dst_ld)[rem] = reinterpret_cast<const float4*>(src + (k * src_block_rows + r) * src_ld)[c];
  }
}

// y[n][oy][ox][c] = relu(max over the 3x3 / stride-2 window (clipped to the map: ceil_mode, no padding) of x + bias[c]):
// the ResNet stem's max pool with the folded-BatchNorm shift and the ReLU applied AFTER the maximum (same values: both
// are monotonic).  One thread = one output pixel x 4 channels; a window row is three 16-byte loads, 256 B per 16 lanes.
__global__ __launch_bounds__(kThreads) void maxpool3s2_bias_relu_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                                       float* __restrict__ y, int H, int W, int OH, int OW, int c4,
                                                                       long total) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
    const int c = (int)(i % c4);
    long t = i / c4;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const long n = t / OH;
    const float4* src = reinterpret_cast<const float4*>(x) + ((n * H + 2 * oy) * W + 2 * ox) * c4 + c;
    const int ny = min(3, H - 2 * oy), nx = min(3, W - 2 * ox);
    float4 m = src[0];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (dy < ny && dx < nx) {
          const float4 v = src[((long)dy * W + dx) * c4];
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    const float4 b = reinterpret_cast<const float4*>(bias)[c];
    m.x = fmaxf(m.x + b.x, 0.f); m.y = fmaxf(m.y + b.y, 0.f); m.z = fmaxf(m.z + b.z, 0.f); m.w = fmaxf(m.w + b.w, 0.f);
    reinterpret_cast<float4*>(y)[i] = m;
  }
}

}  // namespace

extern "C" int dtt_bias_act_nhwc_inplace(float* x, const float* bias, const float* residual, long rows, int channels,
                                         int relu, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(x && bias, "bias_act_nhwc: null pointer");
  DTT_REQUIRE(rows > 0 && channels > 0 && channels % 4 == 0, "bias_act_nhwc: channels must be a positive multiple of 4");
  DTT_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0,
              "bias_act_nhwc: pointers must be 16-byte aligned");
  const long n4 = rows * (channels / 4);
  long blocks = (n4 + kThreads * 4 - 1) / (kThreads * 4);
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks);
  const int c4 = channels / 4;
  if (residual) {
    if (relu) hipLaunchKernelGGL((bias_act_rows_kernel<true, true>), grid, dim3(kThreads), 0, stream, x, bias, residual, n4, c4);
    else hipLaunchKernelGGL((bias_act_rows_kernel<false, true>), grid, dim3(kThreads), 0, stream, x, bias, residual, n4, c4);
  } else {
    if (relu) hipLaunchKernelGGL((bias_act_rows_kernel<true, false>), grid, dim3(kThreads), 0, stream, x, bias, residual, n4, c4);
    else hipLaunchKernelGGL((bias_act_rows_kernel<false, false>), grid, dim3(kThreads), 0, stream, x, bias, residual, n4, c4);
  }
  DTT_CHECK_LAUNCH("bias_act_rows_kernel");
  return 1;
}

extern "C" int dtt_bias_act_inplace(float* x, const float* bias, const float* residual, int batch, int channels,
                                    int hw, int relu, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(x && bias, "bias_act: null pointer");
  DTT_REQUIRE(batch > 0 && channels > 0 && hw > 0, "bias_act: bad shape");
  const long planes = (long)batch * channels;
  // ~8 float4 per thread; planes ride on blockIdx.y
  int bx = (hw / 4 + kThreads * 8 - 1) / (kThreads * 8);
  if (bx < 1) bx = 1;
  dim3 grid(bx, (unsigned)planes);
  DTT_REQUIRE(planes <= 65535, "bias_act: more than 65535 (image, channel) planes in one call");
  if (residual) {
    if (relu) hipLaunchKernelGGL((bias_act_kernel<true, true>), grid, dim3(kThreads), 0, stream, x, bias, residual, channels, hw);
    else hipLaunchKernelGGL((bias_act_kernel<false, true>), grid, dim3(kThreads), 0, stream, x, bias, residual, channels, hw);
  } else {
    if (relu) hipLaunchKernelGGL((bias_act_kernel<true, false>), grid, dim3(kThreads), 0, stream, x, bias, residual, channels, hw);
    else hipLaunchKernelGGL((bias_act_kernel<false, false>), grid, dim3(kThreads), 0, stream, x, bias, residual, channels, hw);
  }
  DTT_CHECK_LAUNCH("bias_act_kernel");
  return 1;
}

extern "C" int dtt_transpose_batched(const float* in, float* out, int batch, int rows, int cols, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(in && out, "transpose_batched: null pointer");
  DTT_REQUIRE(in != out, "transpose_batched: in place is not supported");
  DTT_REQUIRE(batch > 0 && rows > 0 && cols > 0, "transpose_batched: bad shape");
  const dim3 grid((cols + kTile - 1) / kTile, (rows + kTile - 1) / kTile, batch);
  DTT_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "transpose_batched: extent too large");
  const bool vr = cols % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && ((long)rows * cols) % 4 == 0;
  const bool vw = rows % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ((long)rows * cols) % 4 == 0;
  if (vr && vw) hipLaunchKernelGGL((transpose_tiles<true, true>), grid, dim3(kThreads), 0, stream, in, out, rows, cols);
  else if (vr) hipLaunchKernelGGL((transpose_tiles<true, false>), grid, dim3(kThreads), 0, stream, in, out, rows, cols);
  else if (vw) hipLaunchKernelGGL((transpose_tiles<false, true>), grid, dim3(kThreads), 0, stream, in, out, rows, cols);
  else hipLaunchKernelGGL((transpose_tiles<false, false>), grid, dim3(kThreads), 0, stream, in, out, rows, cols);
  DTT_CHECK_LAUNCH("transpose_tiles");
  return 1;
}

namespace {

// ---- many small row-scalings in one launch
// dst_i[r][c] = src_i[r][c] * scale_i[r] for i < n tensors (rows_i x cols_i, dense, row-major in MEMORY order).  The training
// trunk folds the frozen BatchNorm scale of every trainable convolution into its filter each step, forward (w * s) and
// backward (grad * s): ~100 filters of 16 K .. 2.4 M elements.  As one PyTorch multiply each they are ~100 launch-bound
// kernels per direction (torch._foreach_mul takes its per-tensor slow path for broadcast operands); here the pointers ride
// in the kernel arguments, kFoldMax tensors per launch, and a workgroup finds its tensor from the prefix sums of the
// per-tensor workgroup counts.
constexpr int kFoldMax = 48;
constexpr int kFoldElems = kThreads * 16;   // elements per workgroup
struct FoldBatch {
  const float* src[kFoldMax];
  const float* scale[kFoldMax];
  float* dst[kFoldMax];
  int cols[kFoldMax];
  long total[kFoldMax];
  int block0[kFoldMax + 1];
  int n;
};

__global__ __launch_bounds__(kThreads) void scale_rows_batch_kernel(FoldBatch fb) {
  int i = 0;
  while (i + 1 < fb.n && (int)blockIdx.x >= fb.block0[i + 1]) ++i;   // (uniform: scalar loop)
  const float* __restrict__ src = fb.src[i];
  const float* __restrict__ scale = fb.scale[i];
  float* __restrict__ dst = fb.dst[i];
  const int cols = fb.cols[i];
  const long total = fb.total[i];
  const long e0 = (long)((int)blockIdx.x - fb.block0[i]) * kFoldElems;
  const bool vec = (cols & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (vec) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long e = e0 + ((long)u * kThreads + threadIdx.x) * 4;
      if (e < total) {   // (cols % 4 == 0: a float4 never straddles two rows, total % 4 == 0)
        const float sc = scale[e / cols];
        float4 v = *reinterpret_cast<const float4*>(src + e);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        *reinterpret_cast<float4*>(dst + e) = v;
      }
    }
  } else {
    for (int u = 0; u < 16; ++u) {
      const long e = e0 + (long)u * kThreads + threadIdx.x;
      if (e < total) dst[e] = src[e] * scale[e / cols];
    }
  }
}

}  // namespace

extern "C" int dtt_scale_rows_batch(int n, const float* const* src, const float* const* scale, float* const* dst,
                                    const int* rows, const int* cols, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(n >= 0 && (n == 0 || (src && scale && dst && rows && cols)), "scale_rows_batch: null pointer");
  for (int i0 = 0; i0 < n; i0 += kFoldMax) {
    FoldBatch fb;
    fb.n = n - i0 < kFoldMax ? n - i0 : kFoldMax;
    int blocks = 0;
    for (int i = 0; i < fb.n; ++i) {
      DTT_REQUIRE(src[i0 + i] && scale[i0 + i] && dst[i0 + i] && rows[i0 + i] > 0 && cols[i0 + i] > 0,
                  "scale_rows_batch: tensor %d: null pointer or empty shape", i0 + i);
      fb.src[i] = src[i0 + i]; fb.scale[i] = scale[i0 + i]; fb.dst[i] = dst[i0 + i];
      fb.cols[i] = cols[i0 + i];
      fb.total[i] = (long)rows[i0 + i] * cols[i0 + i];
      fb.block0[i] = blocks;
      const long nb = (fb.total[i] + kFoldElems - 1) / kFoldElems;
      DTT_REQUIRE(nb < (1L << 24) && blocks + nb < (1L << 30), "scale_rows_batch: tensor %d too large", i0 + i);
      blocks += (int)nb;
    }
    fb.block0[fb.n] = blocks;
    hipLaunchKernelGGL(scale_rows_batch_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, fb);
    DTT_CHECK_LAUNCH("scale_rows_batch_kernel");
  }
  return 1;
}

// Tracking-head input assembly (rfcn.py:133-140 `torch.cat` of the two legs' box-delta maps, on position-major rows):
// dst[r][k * ncols + c] = src[k * src_block_rows + r][c] for k < n_blocks, r < rows, c < ncols.  dst / src point at the
// first column of interest; leading dimensions in floats; everything 16-byte aligned, ncols % 4 == 0.
extern "C" int dtt_gather_column_blocks(float* dst, long dst_ld, const float* src, long src_ld, long src_block_rows, int n_blocks,
                                        long rows, int ncols, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(dst && src, "gather_column_blocks: null pointer");
  DTT_REQUIRE(rows > 0 && n_blocks > 0 && ncols > 0 && ncols % 4 == 0 && dst_ld % 4 == 0 && src_ld % 4 == 0 &&
              dst_ld >= (long)n_blocks * ncols && src_ld >= ncols && src_block_rows >= rows, "gather_column_blocks: bad shape");
  DTT_REQUIRE(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0,
              "gather_column_blocks: pointers must be 16-byte aligned");
  const long total = rows * n_blocks * (ncols / 4);
  long blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gather_column_blocks_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, dst, dst_ld, src, src_ld,
                     src_block_rows, n_blocks, rows, ncols / 4);
  DTT_CHECK_LAUNCH("gather_column_blocks_kernel");
  return 1;
}

// ResNet stem tail (faster_rcnn/resnet.py:110-117: bn1 -> relu -> MaxPool2d(3, 2, padding 0, ceil_mode)) on a channels-last map
// whose BatchNorm scale is already folded into the convolution: y = relu(maxpool(x) + bias), one pass instead of three.
// x (images, height, width, channels), y (images, out_h, out_w, channels) with out = ceil((size - 3) / 2) + 1.
extern "C" int dtt_maxpool3s2_bias_relu_nhwc(const float* x, const float* bias, float* y, int images, int height, int width,
                                             int channels, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(x && bias && y && x != y, "maxpool_bias_relu: null / aliased pointer");
  DTT_REQUIRE(images > 0 && height >= 3 && width >= 3 && channels > 0 && channels % 4 == 0, "maxpool_bias_relu: bad shape");
  DTT_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
              "maxpool_bias_relu: pointers must be 16-byte aligned");
  const int oh = (height - 3 + 1) / 2 + 1, ow = (width - 3 + 1) / 2 + 1;   // ceil((size - 3) / 2) + 1
  const long total = (long)images * oh * ow * (channels / 4);
  long blocks = (total + kThreads - 1) / kThreads;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(maxpool3s2_bias_relu_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, x, bias, y, height, width, oh, ow,
                     channels / 4, total);
  DTT_CHECK_LAUNCH("maxpool3s2_bias_relu_kernel");
  return 1;
}

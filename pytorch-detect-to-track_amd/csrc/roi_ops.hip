// RoI Align / RoI max-pool / RoI crop (bilinear grid sampler) for gfx950.
//
// Replaces ROIAlignForward/Backward (reference roi_align/src/roi_align_kernel.cu:15-70, 94-143),
// ROIPoolForward/Backward (roi_pooling/src/roi_pooling_kernel.cu:24-93, 128-203) and
// bilinearSamplingFromGrid / backwardBilinearSampling (roi_crop/src/roi_crop_cuda_kernel.cu:47-109, 111-194).
// These ops are write-bound gathers (R*C*P*P outputs from an L2-resident map): one thread per output
// element with the output index fastest so stores coalesce; RoI Align fuses the 2x2 stride-1 pooling
// that RoIAlignAvg / RoIAlignMax run as a second launch (modules/roi_align.py:26-29, 39-42); RoI pool's
// backward is a scatter through argmax instead of the reference's O(pixels x RoIs) gather.
// Arithmetic follows the reference expression by expression (including its double-precision
// sub-expressions) with FP contraction off.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// one bilinear tap of roi_align_kernel.cu:33-68 at sample (ph, pw) of an (ah x aw) grid
__device__ __forceinline__ float align_sample(const float* __restrict__ bottom_data, const float* __restrict__ roi,
                                              float spatial_scale, int c, int ph, int pw, int ah, int aw, int height,
                                              int width, int channels) {
  const float roi_batch_ind = roi[0];
  const float roi_start_w = roi[1] * spatial_scale;
  const float roi_start_h = roi[2] * spatial_scale;
  const float roi_end_w = roi[3] * spatial_scale;
  const float roi_end_h = roi[4] * spatial_scale;
  const float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);
  const float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f);
  const float bin_size_h = (float)((double)roi_height / (ah - 1.));
  const float bin_size_w = (float)((double)roi_width / (aw - 1.));
  const float h = (float)(ph)*bin_size_h + roi_start_h;
  const float w = (float)(pw)*bin_size_w + roi_start_w;
  if (h < 0 || h >= height || w < 0 || w >= width) return 0.f;
  const int hstart = (int)fminf(floorf(h), (float)(height - 2));
  const int wstart = (int)fminf(floorf(w), (float)(width - 2));
  const long img_start = (long)(roi_batch_ind * channels * height * width);
  const float h_ratio = h - (float)(hstart);
  const float w_ratio = w - (float)(wstart);
  const long upleft = img_start + ((long)c * height + hstart) * width + wstart;
  // C++ promotion exactly as roi_align_kernel.cu:65-68 writes it: `data * (1. - h_ratio)` is a double product, `data *
  // h_ratio` is float * float (rounded to float); checked bit for bit against the reference kernel (oracle/_ref)
  const float dl_h = bottom_data[upleft + width] * h_ratio;
  const float dr_hw = (bottom_data[upleft + width + 1] * h_ratio) * w_ratio;
  const double v = (double)bottom_data[upleft] * (1. - h_ratio) * (1. - w_ratio) +
                   (double)bottom_data[upleft + 1] * (1. - h_ratio) * w_ratio + (double)dl_h * (1. - w_ratio) +
                   (double)dr_hw;
  return (float)v;
}

// pool_mode 0: (oh, ow) == sample grid.  1 / 2: samples on (oh+1, ow+1), 2x2 stride-1 avg / max.
__global__ __launch_bounds__(kThreads) void roi_align_fwd(long nthreads, const float* __restrict__ bottom_data,
                                                          float spatial_scale, int height, int width, int channels,
                                                          int oh, int ow, const float* __restrict__ bottom_rois,
                                                          float* __restrict__ top_data, int pool_mode) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int pw = index % ow;
    const int ph = (index / ow) % oh;
    const int c = (index / ow / oh) % channels;
    const int n = index / ow / oh / channels;
    const float* roi = bottom_rois + (long)n * 5;
    float v;
    if (pool_mode == 0) {
      v = align_sample(bottom_data, roi, spatial_scale, c, ph, pw, oh, ow, height, width, channels);
    } else {
      const float s00 = align_sample(bottom_data, roi, spatial_scale, c, ph, pw, oh + 1, ow + 1, height, width, channels);
      const float s01 = align_sample(bottom_data, roi, spatial_scale, c, ph, pw + 1, oh + 1, ow + 1, height, width, channels);
      const float s10 = align_sample(bottom_data, roi, spatial_scale, c, ph + 1, pw, oh + 1, ow + 1, height, width, channels);
      const float s11 = align_sample(bottom_data, roi, spatial_scale, c, ph + 1, pw + 1, oh + 1, ow + 1, height, width, channels);
      v = pool_mode == 1 ? (((s00 + s01) + s10) + s11) / 4.f : fmaxf(fmaxf(s00, s01), fmaxf(s10, s11));
    }
    top_data[index] = v;
  }
}

// Map-stationary RoI Align: one workgroup owns CB channel planes of one image in LDS and walks that image's RoIs.
// The thread-per-output kernel above recomputes the RoI geometry (double-precision divisions) for every sample of
// every channel and gathers its 16 taps from L2; here the geometry and the bilinear weights of an output bin are
// computed once and reused for the CB planes, and the taps come from LDS.  Per-sample arithmetic and the 2x2 pooling
// order are the same expressions as align_sample / roi_align_fwd, so results are bit-identical.
// grid (channels / CB, batch).  LDS: CB planes | RoI list (indices) -- staged in chunks of kAlignChunk RoIs.
constexpr int kAlignCB = 4;
constexpr int kAlignChunk = 1024;
constexpr int kAlignThreads = 1024;   // 4 waves per SIMD: the double-precision tap arithmetic is a long dependent chain

struct AxisTap { int start; float ratio; bool valid; };

__device__ __forceinline__ AxisTap align_axis(float coord, int extent) {
  AxisTap t;
  t.valid = !(coord < 0 || coord >= extent);
  t.start = (int)fminf(floorf(coord), (float)(extent - 2));
  t.ratio = coord - (float)t.start;
  return t;
}

__global__ __launch_bounds__(kAlignThreads) void roi_align_planes(const float* __restrict__ bottom_data, float spatial_scale,
                                                             int num_rois, int height, int width, int channels, int oh,
                                                             int ow, const float* __restrict__ bottom_rois,
                                                             float* __restrict__ top_data, int pool_mode) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  const int hw = height * width;
  float* planes = smem_f;                                        // [kAlignCB][hw]
  int* list = reinterpret_cast<int*>(smem_f + kAlignCB * hw);    // [kAlignChunk]
  __shared__ int list_n;
  const int c0 = blockIdx.x * kAlignCB, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int ncb = min(kAlignCB, channels - c0);
  for (int cb = 0; cb < ncb; ++cb) {
    const float* src = bottom_data + ((long)b * channels + c0 + cb) * hw;
    for (int i = tid; i < hw; i += kAlignThreads) planes[cb * hw + i] = src[i];
  }
  const int ah = pool_mode == 0 ? oh : oh + 1, aw = pool_mode == 0 ? ow : ow + 1;   // sample grid
  const int bins = oh * ow;
  for (int r0 = 0; r0 < num_rois; r0 += kAlignChunk) {
    __syncthreads();
    if (tid == 0) list_n = 0;
    __syncthreads();
    // this image's RoIs of the chunk, in order (ballot compaction, one LDS atomic per wave)
    for (int i0 = 0; i0 < kAlignChunk; i0 += kAlignThreads) {
      const int n = r0 + i0 + tid;
      const bool mine = n < num_rois && bottom_rois[(long)n * 5] == (float)b;
      const unsigned long long mk = __ballot(mine);
      if (mk) {
        int base = 0;
        const int leader = __builtin_ctzll(mk);
        if (lane == leader) base = atomicAdd(&list_n, __builtin_popcountll(mk));
        base = __builtin_amdgcn_readlane(base, leader);
        if (mine) list[base + __builtin_popcountll(mk & ((1ULL << lane) - 1ULL))] = n;
      }
    }
    __syncthreads();
    const int total = list_n * bins;
    for (int idx = tid; idx < total; idx += kAlignThreads) {
      const int n = list[idx / bins], bin = idx % bins;
      const int ph = bin / ow, pw = bin - ph * ow;
      const float* roi = bottom_rois + (long)n * 5;
      // roi_align_kernel.cu:33-46, once per output bin
      const float roi_start_w = roi[1] * spatial_scale;
      const float roi_start_h = roi[2] * spatial_scale;
      const float roi_end_w = roi[3] * spatial_scale;
      const float roi_end_h = roi[4] * spatial_scale;
      const float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);
      const float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f);
      const float bin_size_h = (float)((double)roi_height / (ah - 1.));
      const float bin_size_w = (float)((double)roi_width / (aw - 1.));
      const int ns = pool_mode == 0 ? 1 : 2;
      AxisTap th[2], tw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        th[i] = align_axis((float)(ph + i) * bin_size_h + roi_start_h, height);
        tw[i] = align_axis((float)(pw + i) * bin_size_w + roi_start_w, width);
      }
      for (int cb = 0; cb < ncb; ++cb) {
        const float* pl = planes + cb * hw;
        float s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (i < ns && j < ns && th[i].valid && tw[j].valid) {
              const int ul = th[i].start * width + tw[j].start;
              const float hr = th[i].ratio, wr = tw[j].ratio;
              const float dl_h = pl[ul + width] * hr;  // float products, as in align_value()
              const float dr_hw = (pl[ul + width + 1] * hr) * wr;
              const double v = (double)pl[ul] * (1. - hr) * (1. - wr) + (double)pl[ul + 1] * (1. - hr) * wr +
                               (double)dl_h * (1. - wr) + (double)dr_hw;
              s[i][j] = (float)v;
            }
          }
        float v;
        if (pool_mode == 0) v = s[0][0];
        else if (pool_mode == 1) v = (((s[0][0] + s[0][1]) + s[1][0]) + s[1][1]) / 4.f;
        else v = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[1][0], s[1][1]));
        top_data[(((long)n * channels + c0 + cb) * oh + ph) * ow + pw] = v;
      }
    }
  }
}

// roi_align_kernel.cu:94-143
__global__ __launch_bounds__(kThreads) void roi_align_bwd(long nthreads, const float* __restrict__ top_diff,
                                                          float spatial_scale, int height, int width, int channels,
                                                          int ah, int aw, float* __restrict__ bottom_diff,
                                                          const float* __restrict__ bottom_rois) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int pw = index % aw;
    const int ph = (index / aw) % ah;
    const int c = (index / aw / ah) % channels;
    const int n = index / aw / ah / channels;
    const float* roi = bottom_rois + (long)n * 5;
    const float roi_batch_ind = roi[0];
    const float roi_start_w = roi[1] * spatial_scale;
    const float roi_start_h = roi[2] * spatial_scale;
    const float roi_end_w = roi[3] * spatial_scale;
    const float roi_end_h = roi[4] * spatial_scale;
    const float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.), 0.f);
    const float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.), 0.f);
    const float bin_size_h = (float)((double)roi_height / (ah - 1.));
    const float bin_size_w = (float)((double)roi_width / (aw - 1.));
    const float h = (float)(ph)*bin_size_h + roi_start_h;
    const float w = (float)(pw)*bin_size_w + roi_start_w;
    if (h < 0 || h >= height || w < 0 || w >= width) continue;
    const int hstart = (int)fminf(floorf(h), (float)(height - 2));
    const int wstart = (int)fminf(floorf(w), (float)(width - 2));
    const long img_start = (long)(roi_batch_ind * channels * height * width);
    const float h_ratio = h - (float)(hstart);
    const float w_ratio = w - (float)(wstart);
    const long upleft = img_start + ((long)c * height + hstart) * width + wstart;
    const double g = (double)top_diff[index];
    atomicAdd(bottom_diff + upleft, (float)(g * (1. - h_ratio) * (1 - w_ratio)));
    atomicAdd(bottom_diff + upleft + 1, (float)(g * (1. - h_ratio) * w_ratio));
    // roi_align_kernel.cu:140-141: float * float * float, no double operand in these two
    atomicAdd(bottom_diff + upleft + width, (top_diff[index] * h_ratio) * (1 - w_ratio));
    atomicAdd(bottom_diff + upleft + width + 1, (top_diff[index] * h_ratio) * w_ratio);
  }
}

// roi_pooling_kernel.cu:24-93
__global__ __launch_bounds__(kThreads) void roi_pool_fwd(long nthreads, const float* __restrict__ bottom_data,
                                                         float spatial_scale, int height, int width, int channels,
                                                         int pooled_height, int pooled_width,
                                                         const float* __restrict__ bottom_rois,
                                                         float* __restrict__ top_data, int* __restrict__ argmax_data) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int c = (index / pooled_width / pooled_height) % channels;
    const int n = index / pooled_width / pooled_height / channels;
    const float* roi = bottom_rois + (long)n * 5;
    const int roi_batch_ind = (int)roi[0];
    const int roi_start_w = (int)roundf(roi[1] * spatial_scale);
    const int roi_start_h = (int)roundf(roi[2] * spatial_scale);
    const int roi_end_w = (int)roundf(roi[3] * spatial_scale);
    const int roi_end_h = (int)roundf(roi[4] * spatial_scale);
    const int roi_width = (int)fmaxf((float)(roi_end_w - roi_start_w + 1), 1.f);
    const int roi_height = (int)fmaxf((float)(roi_end_h - roi_start_h + 1), 1.f);
    const float bin_size_h = (float)(roi_height) / (float)(pooled_height);
    const float bin_size_w = (float)(roi_width) / (float)(pooled_width);
    int hstart = (int)(floorf((float)(ph)*bin_size_h));
    int wstart = (int)(floorf((float)(pw)*bin_size_w));
    int hend = (int)(ceilf((float)(ph + 1) * bin_size_h));
    int wend = (int)(ceilf((float)(pw + 1) * bin_size_w));
    hstart = (int)fminf(fmaxf((float)(hstart + roi_start_h), 0.f), (float)height);
    hend = (int)fminf(fmaxf((float)(hend + roi_start_h), 0.f), (float)height);
    wstart = (int)fminf(fmaxf((float)(wstart + roi_start_w), 0.f), (float)width);
    wend = (int)fminf(fmaxf((float)(wend + roi_start_w), 0.f), (float)width);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    float maxval = is_empty ? 0 : -3.402823466e+38F;
    int maxidx = -1;
    const int off = (roi_batch_ind * channels + c) * height * width;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) {
        const int bottom_index = h * width + w;
        const float v = bottom_data[off + bottom_index];
        if (v > maxval) { maxval = v; maxidx = off + bottom_index; }
      }
    top_data[index] = maxval;
    if (argmax_data) argmax_data[index] = maxidx;
  }
}

// Scatter form of roi_pooling_kernel.cu:128-203.  The reference gathers: every bottom element visits every
// RoI that contains it and every pooled unit that could have pooled it, adding top_diff where argmax matches.
// Here each pooled unit scatters to its argmax, after re-applying the reference's two admission tests
// (element inside the rounded RoI, pooled unit inside the feasible [phstart, phend) x [pwstart, pwend) window),
// so malformed RoIs drop their gradient exactly as the reference does.
__global__ __launch_bounds__(kThreads) void roi_pool_bwd(long nthreads, const float* __restrict__ top_diff,
                                                         const int* __restrict__ argmax_data, float spatial_scale,
                                                         int height, int width, int channels, int pooled_height,
                                                         int pooled_width, const float* __restrict__ bottom_rois,
                                                         float* __restrict__ bottom_diff) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int a = argmax_data[index];
    if (a < 0) continue;
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int n = index / pooled_width / pooled_height / channels;
    const int w = a % width, h = (a / width) % height;
    const float* roi = bottom_rois + (long)n * 5;
    const int roi_start_w = (int)roundf(roi[1] * spatial_scale);
    const int roi_start_h = (int)roundf(roi[2] * spatial_scale);
    const int roi_end_w = (int)roundf(roi[3] * spatial_scale);
    const int roi_end_h = (int)roundf(roi[4] * spatial_scale);
    if (!(w >= roi_start_w && w <= roi_end_w && h >= roi_start_h && h <= roi_end_h)) continue;
    const int roi_width = (int)fmaxf((float)(roi_end_w - roi_start_w + 1), 1.f);
    const int roi_height = (int)fmaxf((float)(roi_end_h - roi_start_h + 1), 1.f);
    const float bin_size_h = (float)(roi_height) / (float)(pooled_height);
    const float bin_size_w = (float)(roi_width) / (float)(pooled_width);
    int phstart = (int)floorf((float)(h - roi_start_h) / bin_size_h);
    int phend = (int)ceilf((float)(h - roi_start_h + 1) / bin_size_h);
    int pwstart = (int)floorf((float)(w - roi_start_w) / bin_size_w);
    int pwend = (int)ceilf((float)(w - roi_start_w + 1) / bin_size_w);
    phstart = (int)fminf(fmaxf((float)phstart, 0.f), (float)pooled_height);
    phend = (int)fminf(fmaxf((float)phend, 0.f), (float)pooled_height);
    pwstart = (int)fminf(fmaxf((float)pwstart, 0.f), (float)pooled_width);
    pwend = (int)fminf(fmaxf((float)pwend, 0.f), (float)pooled_width);
    if (ph < phstart || ph >= phend || pw < pwstart || pw >= pwend) continue;
    atomicAdd(bottom_diff + a, top_diff[index]);
  }
}

// roi_crop_cuda_kernel.cu:11-22
__device__ __forceinline__ void get_top_left(float x, int width, int& point, float& weight) {
  const float xcoord = (x + 1) * (width - 1) / 2;
  point = (int)floorf(xcoord);
  weight = 1 - (xcoord - point);
}
__device__ __forceinline__ bool between(int v, int lo, int hi) { return v >= lo && v <= hi; }

// roi_crop_cuda_kernel.cu:47-109 for contiguous BCHW images / (B, H, W, 2) (y, x) grids
__global__ __launch_bounds__(kThreads) void roi_crop_fwd(long nthreads, const float* __restrict__ images,
                                                         const float* __restrict__ grids, float* __restrict__ output,
                                                         int ic, int ih, int iw, int oh, int ow, int roi_per_image) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int xOut = index % ow;
    const int yOut = (index / ow) % oh;
    const int cOut = (index / ow / oh) % ic;
    const int b = index / ow / oh / ic;
    const int b_input = b / roi_per_image;
    const float yf = grids[(((long)b * oh + yOut) * ow + xOut) * 2 + 0];
    const float xf = grids[(((long)b * oh + yOut) * ow + xOut) * 2 + 1];
    int yInTopLeft, xInTopLeft;
    float yWeightTopLeft, xWeightTopLeft;
    get_top_left(xf, iw, xInTopLeft, xWeightTopLeft);
    get_top_left(yf, ih, yInTopLeft, yWeightTopLeft);
    const long tl = (((long)b_input * ic + cOut) * ih + yInTopLeft) * iw + xInTopLeft;
    const bool tlIn = between(xInTopLeft, 0, iw - 1) && between(yInTopLeft, 0, ih - 1);
    const bool trIn = between(xInTopLeft + 1, 0, iw - 1) && between(yInTopLeft, 0, ih - 1);
    const bool blIn = between(xInTopLeft, 0, iw - 1) && between(yInTopLeft + 1, 0, ih - 1);
    const bool brIn = between(xInTopLeft + 1, 0, iw - 1) && between(yInTopLeft + 1, 0, ih - 1);
    float v = 0.f;
    if (tlIn || trIn || blIn || brIn) {
      const float inTopLeft = tlIn ? images[tl] : 0.f;
      const float inTopRight = trIn ? images[tl + 1] : 0.f;
      const float inBottomLeft = blIn ? images[tl + iw] : 0.f;
      const float inBottomRight = brIn ? images[tl + iw + 1] : 0.f;
      v = xWeightTopLeft * yWeightTopLeft * inTopLeft + (1 - xWeightTopLeft) * yWeightTopLeft * inTopRight +
          xWeightTopLeft * (1 - yWeightTopLeft) * inBottomLeft +
          (1 - xWeightTopLeft) * (1 - yWeightTopLeft) * inBottomRight;
    }
    output[index] = v;
  }
}

// roi_crop_cuda_kernel.cu:111-194 (image gradient only; the grid gradient is never written there)
__global__ __launch_bounds__(kThreads) void roi_crop_bwd(long nthreads, const float* __restrict__ grids,
                                                         float* __restrict__ grad_images,
                                                         const float* __restrict__ grad_output, int ic, int ih,
                                                         int iw, int oh, int ow, int roi_per_image) {
  for (long index = (long)blockIdx.x * kThreads + threadIdx.x; index < nthreads; index += (long)gridDim.x * kThreads) {
    const int xOut = index % ow;
    const int yOut = (index / ow) % oh;
    const int cOut = (index / ow / oh) % ic;
    const int b = index / ow / oh / ic;
    const int b_input = b / roi_per_image;
    const float yf = grids[(((long)b * oh + yOut) * ow + xOut) * 2 + 0];
    const float xf = grids[(((long)b * oh + yOut) * ow + xOut) * 2 + 1];
    int yInTopLeft, xInTopLeft;
    float yWeightTopLeft, xWeightTopLeft;
    get_top_left(xf, iw, xInTopLeft, xWeightTopLeft);
    get_top_left(yf, ih, yInTopLeft, yWeightTopLeft);
    const long tl = (((long)b_input * ic + cOut) * ih + yInTopLeft) * iw + xInTopLeft;
    const bool tlIn = between(xInTopLeft, 0, iw - 1) && between(yInTopLeft, 0, ih - 1);
    const bool trIn = between(xInTopLeft + 1, 0, iw - 1) && between(yInTopLeft, 0, ih - 1);
    const bool blIn = between(xInTopLeft, 0, iw - 1) && between(yInTopLeft + 1, 0, ih - 1);
    const bool brIn = between(xInTopLeft + 1, 0, iw - 1) && between(yInTopLeft + 1, 0, ih - 1);
    const float g = grad_output[index];
    if (tlIn) atomicAdd(grad_images + tl, xWeightTopLeft * yWeightTopLeft * g);
    if (trIn) atomicAdd(grad_images + tl + 1, (1 - xWeightTopLeft) * yWeightTopLeft * g);
    if (blIn) atomicAdd(grad_images + tl + iw, xWeightTopLeft * (1 - yWeightTopLeft) * g);
    if (brIn) atomicAdd(grad_images + tl + iw + 1, (1 - xWeightTopLeft) * (1 - yWeightTopLeft) * g);
  }
}

inline int grid_for(long n) { return (int)((n + kThreads - 1) / kThreads > 262144 ? 262144 : (n + kThreads - 1) / kThreads); }

}  // namespace

extern "C" int dtt_roi_align_forward(const float* bottom_data, float spatial_scale, int num_rois, int height,
                                     int width, int channels, int aligned_height, int aligned_width,
                                     const float* bottom_rois, float* top_data, int pool_mode, void* stream) {
  DTT_REQUIRE(num_rois >= 0 && height > 1 && width > 1 && channels > 0, "roi_align forward: bad shape");
  DTT_REQUIRE(pool_mode >= 0 && pool_mode <= 2, "roi_align forward: pool_mode must be 0, 1 or 2");
  DTT_REQUIRE(aligned_height > (pool_mode == 0 ? 1 : 0) && aligned_width > (pool_mode == 0 ? 1 : 0),
              "roi_align forward: aligned size too small");
  const long n = (long)num_rois * channels * aligned_height * aligned_width;
  if (n == 0) return 1;
  DTT_REQUIRE(bottom_data && bottom_rois && top_data, "roi_align forward: null pointer");
  hipLaunchKernelGGL(roi_align_fwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n,
                     bottom_data, spatial_scale, height, width, channels, aligned_height, aligned_width, bottom_rois,
                     top_data, pool_mode);
  DTT_CHECK_LAUNCH("roi_align_fwd");
  return 1;
}

extern "C" int dtt_roi_align_forward_planes(const float* bottom_data, float spatial_scale, int batch_size, int num_rois,
                                            int height, int width, int channels, int aligned_height,
                                            int aligned_width, const float* bottom_rois, float* top_data,
                                            int pool_mode, void* stream) {
  DTT_REQUIRE(batch_size > 0, "roi_align forward: bad batch size");
  DTT_REQUIRE(num_rois >= 0 && height > 1 && width > 1 && channels > 0, "roi_align forward: bad shape");
  DTT_REQUIRE(pool_mode >= 0 && pool_mode <= 2, "roi_align forward: pool_mode must be 0, 1 or 2");
  DTT_REQUIRE(aligned_height > (pool_mode == 0 ? 1 : 0) && aligned_width > (pool_mode == 0 ? 1 : 0),
              "roi_align forward: aligned size too small");
  if ((long)num_rois * channels * aligned_height * aligned_width == 0) return 1;
  DTT_REQUIRE(bottom_data && bottom_rois && top_data, "roi_align forward: null pointer");
  const size_t lds = ((size_t)kAlignCB * height * width + kAlignChunk) * sizeof(float);
  if (lds > 150 * 1024)   // planes do not fit LDS: thread-per-output path
    return dtt_roi_align_forward(bottom_data, spatial_scale, num_rois, height, width, channels, aligned_height,
                                 aligned_width, bottom_rois, top_data, pool_mode, stream);
  static DttDeviceOnce attr_once;
  bool& attr = attr_once.here();   // the attribute is per device, not per process
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(roi_align_planes),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);  // + static list_n
    DTT_REQUIRE(e == hipSuccess, "roi_align: cannot raise dynamic LDS limit");
    attr = true;
  }
  hipLaunchKernelGGL(roi_align_planes, dim3((channels + kAlignCB - 1) / kAlignCB, batch_size), dim3(kAlignThreads), lds,
                     static_cast<hipStream_t>(stream), bottom_data, spatial_scale, num_rois, height, width, channels,
                     aligned_height, aligned_width, bottom_rois, top_data, pool_mode);
  DTT_CHECK_LAUNCH("roi_align_planes");
  return 1;
}

extern "C" int dtt_roi_align_backward(const float* top_diff, float spatial_scale, int batch_size, int num_rois,
                                      int height, int width, int channels, int aligned_height, int aligned_width,
                                      const float* bottom_rois, float* bottom_diff, void* stream) {
  (void)batch_size;
  DTT_REQUIRE(num_rois >= 0 && height > 1 && width > 1 && channels > 0 && aligned_height > 1 && aligned_width > 1,
              "roi_align backward: bad shape");
  const long n = (long)num_rois * channels * aligned_height * aligned_width;
  if (n == 0) return 1;
  DTT_REQUIRE(top_diff && bottom_rois && bottom_diff, "roi_align backward: null pointer");
  hipLaunchKernelGGL(roi_align_bwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n,
                     top_diff, spatial_scale, height, width, channels, aligned_height, aligned_width, bottom_diff,
                     bottom_rois);
  DTT_CHECK_LAUNCH("roi_align_bwd");
  return 1;
}

extern "C" int dtt_roi_pool_forward(const float* bottom_data, float spatial_scale, int num_rois, int height,
                                    int width, int channels, int pooled_height, int pooled_width,
                                    const float* bottom_rois, float* top_data, int* argmax_data, void* stream) {
  DTT_REQUIRE(num_rois >= 0 && height > 0 && width > 0 && channels > 0 && pooled_height > 0 && pooled_width > 0,
              "roi_pool forward: bad shape");
  const long n = (long)num_rois * channels * pooled_height * pooled_width;
  if (n == 0) return 1;
  DTT_REQUIRE(bottom_data && bottom_rois && top_data, "roi_pool forward: null pointer");
  hipLaunchKernelGGL(roi_pool_fwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n,
                     bottom_data, spatial_scale, height, width, channels, pooled_height, pooled_width, bottom_rois,
                     top_data, argmax_data);
  DTT_CHECK_LAUNCH("roi_pool_fwd");
  return 1;
}

extern "C" int dtt_roi_pool_backward(const float* top_diff, float spatial_scale, int batch_size, int num_rois,
                                     int height, int width, int channels, int pooled_height, int pooled_width,
                                     const float* bottom_rois, float* bottom_diff, const int* argmax_data,
                                     void* stream) {
  (void)batch_size;
  const long n = (long)num_rois * channels * pooled_height * pooled_width;
  if (n == 0) return 1;
  DTT_REQUIRE(top_diff && bottom_diff && argmax_data && bottom_rois, "roi_pool backward: null pointer");
  hipLaunchKernelGGL(roi_pool_bwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n,
                     top_diff, argmax_data, spatial_scale, height, width, channels, pooled_height, pooled_width,
                     bottom_rois, bottom_diff);
  DTT_CHECK_LAUNCH("roi_pool_bwd");
  return 1;
}

extern "C" int dtt_roi_crop_forward(int oc, int ow, int oh, int ob, int ic, int ih, int iw, int ib,
                                    const float* inputImages, const float* grids, float* output, void* stream) {
  DTT_REQUIRE(oc == ic, "roi_crop forward: output channels (%d) != input channels (%d)", oc, ic);
  DTT_REQUIRE(ib > 0 && ob >= 0 && ob % ib == 0, "roi_crop forward: %d RoIs not a multiple of %d images", ob, ib);
  const long n = (long)ob * oc * oh * ow;
  if (n == 0) return 1;
  DTT_REQUIRE(inputImages && grids && output, "roi_crop forward: null pointer");
  hipLaunchKernelGGL(roi_crop_fwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n,
                     inputImages, grids, output, ic, ih, iw, oh, ow, ob / ib);
  DTT_CHECK_LAUNCH("roi_crop_fwd");
  return 1;
}

extern "C" int dtt_roi_crop_backward(int goc, int gow, int goh, int gob, int ic, int ih, int iw, int ib,
                                     const float* inputImages, const float* grids, float* gradInputImages,
                                     const float* gradOutput, void* stream) {
  (void)inputImages;
  DTT_REQUIRE(goc == ic, "roi_crop backward: channel mismatch");
  DTT_REQUIRE(ib > 0 && gob >= 0 && gob % ib == 0, "roi_crop backward: %d RoIs not a multiple of %d images", gob, ib);
  const long n = (long)gob * goc * goh * gow;
  if (n == 0) return 1;
  DTT_REQUIRE(grids && gradInputImages && gradOutput, "roi_crop backward: null pointer");
  hipLaunchKernelGGL(roi_crop_bwd, dim3(grid_for(n)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), n, grids,
                     gradInputImages, gradOutput, ic, ih, iw, goh, gow, gob / ib);
  DTT_CHECK_LAUNCH("roi_crop_bwd");
  return 1;
}

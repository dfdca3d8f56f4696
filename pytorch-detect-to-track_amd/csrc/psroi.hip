// Position-sensitive RoI pooling for gfx950.
//
// Replaces PSROIPoolForward / PSROIPoolBackward (reference psroi_pooling/src/psroi_pooling_kernel.cu:15-79,
// 109-170).  The reference maps one thread to one output bin, so neighbouring threads walk different
// channels (stride H*W) and every RoI re-reads the score map.  Here the map is the stationary operand:
// one workgroup owns one (image, channel) score plane = one (ctop, ph, pw) bin position, stages the
// plane in LDS with coalesced loads (each plane is read from HBM exactly once), and its threads walk the
// RoIs of that image, summing their bin out of LDS in the reference's (h, w) order.  The backward does
// the transpose: bins are scattered into the LDS plane with ds_add_f32 and the finished plane is written
// once, coalesced -- no global atomics, no pre-zeroed output.
//
// Bin-edge arithmetic follows the reference operation by operation (double where the reference's
// literals promote to double, no FMA contraction), so bin boundaries match the oracle bit for bit.
#include "common.h"
#include "psroi_bin.h"

namespace {

constexpr int kThreads = 256;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#ifndef DTT_PSROI_ABLATE
#define DTT_PSROI_ABLATE 0   // developer timing experiments: 1 = staging only, 2 = no plane staging, 3 = no bin geometry
#endif

// Stage one contiguous plane (hw floats at 4-byte alignment) into LDS.  All of a thread's 16-byte loads are issued
// before the first LDS write: a plain `plane[i] = src[i]` loop waits out one global-load latency per element
// (10 per thread at 38 x 67), which is what used to bound the whole kernel.
__device__ __forceinline__ void stage_plane(float* __restrict__ plane, const float* __restrict__ src, int hw) {
  const int hw4 = hw >> 2;
  constexpr int U = 4;
  const int nt = (int)blockDim.x;
  for (int base = 0; base < hw4; base += nt * U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nt + (int)threadIdx.x;
      if (idx < hw4) __builtin_memcpy(&v[u], src + 4 * idx, 16);   // global_load_dwordx4 at 4-byte alignment
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nt + (int)threadIdx.x;
      if (idx < hw4) *reinterpret_cast<f32x4_t*>(plane + 4 * idx) = v[u];
    }
  }
  const int t = (hw4 << 2) + (int)threadIdx.x;
  if (t < hw) plane[t] = src[t];
}

// grid (channels, batch).  LDS: height*width floats.  TR: write the bins channel-major, top[c][roi] -- every store
// instruction covers consecutive RoIs of one plane (coalesced) instead of one dword per 6 KB; only the fused vote
// reads that layout.
template <bool TR>
__global__ __launch_bounds__(kThreads) void psroi_fwd_plane(
    const float* __restrict__ bottom_data, float spatial_scale, int num_rois, int height, int width, int channels,
    int pooled_height, int pooled_width, const float* __restrict__ bottom_rois, int group_size, int output_dim,
    float* __restrict__ top_data, int* __restrict__ mapping_channel) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  const int c = blockIdx.x, b = blockIdx.y;
  const int gw = c % group_size, gh = (c / group_size) % group_size, ctop = c / (group_size * group_size);
  if (gw >= pooled_width || gh >= pooled_height || ctop >= output_dim) return;  // channel feeds no bin
  const int hw = height * width;
  const float* src = bottom_data + ((long)b * channels + c) * hw;
  // RoI rows for the first kPre rounds of the loop are requested BEFORE the plane is staged and consumed after the
  // barrier: a wave's critical path is then one global round trip (plane and RoI rows together), not one for the
  // plane, one for the batch index and one for the coordinates of every round -- and with ~1.5 workgroup waves per
  // launch the kernel's duration is a small multiple of exactly that path.
  constexpr int kPre = 5;   // 1280 RoIs: the inference shape is 4 images x 300
  float pre[kPre][5];
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int n = (int)threadIdx.x + it * kThreads;
    const float* roi = bottom_rois + (long)min(n, num_rois - 1) * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) pre[it][q] = roi[q];
  }
#if DTT_PSROI_ABLATE != 2
  stage_plane(plane, src, hw);
#endif
  __syncthreads();
#if DTT_PSROI_ABLATE == 1
  if (num_rois > 0) return;
#endif
  auto pool = [&](const float* roi, int n) {
    const Bin bin = psroi_bin(roi, spatial_scale, gh, gw, pooled_height, pooled_width, height, width);
    float out_sum = 0;
    for (int h = bin.hstart; h < bin.hend; ++h)
      for (int w = bin.wstart; w < bin.wend; ++w) out_sum += plane[h * width + w];
    const float bin_area = (float)((bin.hend - bin.hstart) * (bin.wend - bin.wstart));
    const long index = TR ? (long)c * num_rois + n : (((long)n * output_dim + ctop) * pooled_height + gh) * pooled_width + gw;
    top_data[index] = bin.empty ? 0.f : out_sum / bin_area;
    if (!TR && mapping_channel) mapping_channel[index] = c;
  };
#pragma unroll
  for (int it = 0; it < kPre; ++it) {
    const int n = (int)threadIdx.x + it * kThreads;
    if (n < num_rois && (int)pre[it][0] == b) pool(pre[it], n);
  }
  for (int n = (int)threadIdx.x + kPre * kThreads; n < num_rois; n += kThreads) {
    const float* roi = bottom_rois + (long)n * 5;
    if ((int)roi[0] != b) continue;
    pool(roi, n);
  }
}

// grid (channels, batch).  LDS: height*width floats.  Writes the whole bottom_diff plane.
__global__ __launch_bounds__(kThreads) void psroi_bwd_plane(
    const float* __restrict__ top_diff, float spatial_scale, int num_rois, int height, int width, int channels,
    int pooled_height, int pooled_width, const float* __restrict__ bottom_rois, int group_size, int output_dim,
    float* __restrict__ bottom_diff) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  const int c = blockIdx.x, b = blockIdx.y;
  const int hw = height * width;
  float* dst = bottom_diff + ((long)b * channels + c) * hw;
  const int gw = c % group_size, gh = (c / group_size) % group_size, ctop = c / (group_size * group_size);
  const bool used = !(gw >= pooled_width || gh >= pooled_height || ctop >= output_dim);
  for (int i = threadIdx.x; i < hw; i += kThreads) plane[i] = 0.f;
  __syncthreads();
  if (used) {
    for (int n = threadIdx.x; n < num_rois; n += kThreads) {
      const float* roi = bottom_rois + (long)n * 5;
      if ((int)roi[0] != b) continue;
      const Bin bin = psroi_bin(roi, spatial_scale, gh, gw, pooled_height, pooled_width, height, width);
      if (bin.empty) continue;
      const float bin_area = (float)((bin.hend - bin.hstart) * (bin.wend - bin.wstart));
      const long index = (((long)n * output_dim + ctop) * pooled_height + gh) * pooled_width + gw;
      const float diff_val = top_diff[index] / bin_area;
      for (int h = bin.hstart; h < bin.hend; ++h)
        for (int w = bin.wstart; w < bin.wend; ++w) atomicAdd(&plane[h * width + w], diff_val);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hw; i += kThreads) dst[i] = plane[i];
}

// Fallbacks for score planes that do not fit LDS or pooled size > group size: one thread per output bin,
// reference-style (psroi_pooling_kernel.cu:15-79, 109-170).
__global__ void psroi_fwd_generic(long nthreads, const float* __restrict__ bottom_data, float spatial_scale,
                                  int height, int width, int channels, int pooled_height, int pooled_width,
                                  int group_size, int output_dim, const float* __restrict__ bottom_rois,
                                  float* __restrict__ top_data, int* __restrict__ mapping_channel) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads; index += (long)blockDim.x * gridDim.x) {
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int ctop = (index / pooled_width / pooled_height) % output_dim;
    const int n = index / pooled_width / pooled_height / output_dim;
    const float* roi = bottom_rois + (long)n * 5;
    const int roi_batch_ind = (int)roi[0];
    const Bin bin = psroi_bin(roi, spatial_scale, ph, pw, pooled_height, pooled_width, height, width);
    const int c = (ctop * group_size + ph) * group_size + pw;
    const float* src = bottom_data + ((long)roi_batch_ind * channels + c) * height * width;
    float out_sum = 0;
    for (int h = bin.hstart; h < bin.hend; ++h)
      for (int w = bin.wstart; w < bin.wend; ++w) out_sum += src[h * width + w];
    const float bin_area = (float)((bin.hend - bin.hstart) * (bin.wend - bin.wstart));
    top_data[index] = bin.empty ? 0.f : out_sum / bin_area;
    if (mapping_channel) mapping_channel[index] = c;
  }
}

__global__ void psroi_bwd_generic(long nthreads, const float* __restrict__ top_diff,
                                  const int* __restrict__ mapping_channel, float spatial_scale, int height, int width,
                                  int channels, int pooled_height, int pooled_width, int group_size, int output_dim,
                                  float* __restrict__ bottom_diff, const float* __restrict__ bottom_rois) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < nthreads; index += (long)blockDim.x * gridDim.x) {
    const int pw = index % pooled_width;
    const int ph = (index / pooled_width) % pooled_height;
    const int ctop = (index / pooled_width / pooled_height) % output_dim;
    const int n = index / pooled_width / pooled_height / output_dim;
    const float* roi = bottom_rois + (long)n * 5;
    const int roi_batch_ind = (int)roi[0];
    const Bin bin = psroi_bin(roi, spatial_scale, ph, pw, pooled_height, pooled_width, height, width);
    if (bin.empty) continue;
    const int c = mapping_channel ? mapping_channel[index] : (ctop * group_size + ph) * group_size + pw;
    float* dst = bottom_diff + ((long)roi_batch_ind * channels + c) * height * width;
    const float bin_area = (float)((bin.hend - bin.hstart) * (bin.wend - bin.wstart));
    const float diff_val = top_diff[index] / bin_area;
    for (int h = bin.hstart; h < bin.hend; ++h)
      for (int w = bin.wstart; w < bin.wend; ++w) atomicAdd(dst + h * width + w, diff_val);
  }
}

// 7x7 vote = nn.AvgPool2d((7,7)) over the pooled bins (rfcn.py:62-64): one thread per (roi, ctop),
// row-major sum then one division by the bin count.
__global__ void psroi_vote(const float* __restrict__ top_data, long n_out, int bins, float* __restrict__ vote) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const float* p = top_data + i * bins;
  float s = 0.f;
  for (int k = 0; k < bins; ++k) s += p[k];
  vote[i] = s / (float)bins;
}

// The same vote over the channel-major scratch of psroi_fwd_plane<true>: thread = (ctop, roi) with the RoI fastest, so
// the 49 reads of a wave are 49 coalesced rows.  Same summation order (ph, pw row-major) and the same division.
__global__ void psroi_vote_tr(const float* __restrict__ top_t, int num_rois, int output_dim, int group_size,
                              int pooled_height, int pooled_width, float* __restrict__ vote) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)num_rois * output_dim) return;
  const int n = (int)(i % num_rois), ctop = (int)(i / num_rois);
  float s = 0.f;
  if (pooled_height == 7 && pooled_width == 7 && group_size == 7) {   // the R-FCN shape: 49 independent coalesced loads
    const float* p = top_t + (long)ctop * 49 * num_rois + n;
    float v[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) v[k] = p[(long)k * num_rois];
#pragma unroll
    for (int k = 0; k < 49; ++k) s += v[k];
  } else {
    for (int ph = 0; ph < pooled_height; ++ph)
      for (int pw = 0; pw < pooled_width; ++pw)
        s += top_t[(long)((ctop * group_size + ph) * group_size + pw) * num_rois + n];
  }
  vote[(long)n * output_dim + ctop] = s / (float)(pooled_height * pooled_width);
}

size_t fwd_plane_lds(int height, int width, int num_rois) {
  (void)num_rois;
  return (((size_t)height * width + 3) & ~(size_t)3) * sizeof(float);
}

bool plane_path_ok(int height, int width, int pooled_height, int pooled_width, int group_size) {
  return (size_t)height * width * sizeof(float) <= 120 * 1024 && pooled_height <= group_size &&
         pooled_width <= group_size;
}

int raise_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return 1;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { dtt_set_error("psroi: cannot raise dynamic LDS limit: %s", hipGetErrorString(e)); return 0; }
  return 1;
}

int check_common(int batch_size, int num_rois, int height, int width, int channels, int pooled_height,
                 int pooled_width, int group_size, int output_dim) {
  DTT_REQUIRE(batch_size > 0 && height > 0 && width > 0 && channels > 0, "psroi: bad feature shape");
  DTT_REQUIRE(num_rois >= 0, "psroi: negative num_rois");
  DTT_REQUIRE(pooled_height > 0 && pooled_width > 0 && group_size > 0 && output_dim > 0, "psroi: bad pooling parameters");
  DTT_REQUIRE((long)((long)(output_dim - 1) * group_size + (pooled_height - 1)) * group_size + (pooled_width - 1) < channels,
              "psroi: output_dim*group_size^2 exceeds the %d input channels", channels);
  return 1;
}

}  // namespace

extern "C" int dtt_psroi_pool_forward(const float* bottom_data, float spatial_scale, int batch_size, int num_rois,
                                      int height, int width, int channels, int pooled_height, int pooled_width,
                                      const float* bottom_rois, int group_size, int output_dim, float* top_data,
                                      int* mapping_channel, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!check_common(batch_size, num_rois, height, width, channels, pooled_height, pooled_width, group_size, output_dim))
    return 0;
  if (num_rois == 0) return 1;
  DTT_REQUIRE(bottom_data && bottom_rois && top_data, "psroi forward: null pointer");
  if (plane_path_ok(height, width, pooled_height, pooled_width, group_size)) {
    const size_t lds = fwd_plane_lds(height, width, num_rois);
    if (!raise_lds(reinterpret_cast<const void*>(psroi_fwd_plane<false>), lds)) return 0;
    dtt_prof_begin("psroi_fwd_plane", stream);
    hipLaunchKernelGGL(psroi_fwd_plane<false>, dim3(channels, batch_size), dim3(kThreads), lds, stream, bottom_data,
                       spatial_scale, num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois,
                       group_size, output_dim, top_data, mapping_channel);
    dtt_prof_end("psroi_fwd_plane", stream);
  } else {
    const long n = (long)num_rois * output_dim * pooled_height * pooled_width;
    hipLaunchKernelGGL(psroi_fwd_generic, dim3(min(dtt_cdiv(n, 256), 65535)), dim3(256), 0, stream, n, bottom_data,
                       spatial_scale, height, width, channels, pooled_height, pooled_width, group_size, output_dim,
                       bottom_rois, top_data, mapping_channel);
  }
  DTT_CHECK_LAUNCH("psroi forward");
  return 1;
}

extern "C" int dtt_psroi_pool_vote_forward(const float* bottom_data, float spatial_scale, int batch_size,
                                           int num_rois, int height, int width, int channels, int pooled_height,
                                           int pooled_width, const float* bottom_rois, int group_size,
                                           int output_dim, float* top_data, float* vote_out, void* stream_) {
  DTT_REQUIRE(top_data && vote_out, "psroi vote: top_data scratch and vote_out are required");
  if (!dtt_psroi_pool_forward(bottom_data, spatial_scale, batch_size, num_rois, height, width, channels,
                              pooled_height, pooled_width, bottom_rois, group_size, output_dim, top_data, nullptr,
                              stream_))
    return 0;
  const long n_out = (long)num_rois * output_dim;
  if (n_out == 0) return 1;
  hipLaunchKernelGGL(psroi_vote, dim3(dtt_cdiv(n_out, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                     top_data, n_out, pooled_height * pooled_width, vote_out);
  DTT_CHECK_LAUNCH("psroi vote");
  return 1;
}

extern "C" int dtt_psroi_vote_forward(const float* bottom_data, float spatial_scale, int batch_size, int num_rois,
                                      int height, int width, int channels, int pooled_height, int pooled_width,
                                      const float* bottom_rois, int group_size, int output_dim, float* scratch,
                                      float* vote_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!check_common(batch_size, num_rois, height, width, channels, pooled_height, pooled_width, group_size, output_dim))
    return 0;
  if (num_rois == 0) return 1;
  DTT_REQUIRE(bottom_data && bottom_rois && scratch && vote_out, "psroi vote: null pointer");
  if (!plane_path_ok(height, width, pooled_height, pooled_width, group_size))   // planes too large for LDS: two-step path
    return dtt_psroi_pool_vote_forward(bottom_data, spatial_scale, batch_size, num_rois, height, width, channels,
                                       pooled_height, pooled_width, bottom_rois, group_size, output_dim, scratch,
                                       vote_out, stream_);
  const size_t lds = fwd_plane_lds(height, width, num_rois);
  if (!raise_lds(reinterpret_cast<const void*>(psroi_fwd_plane<true>), lds)) return 0;
  dtt_prof_begin("psroi_fwd_plane", stream);
  hipLaunchKernelGGL(psroi_fwd_plane<true>, dim3(channels, batch_size), dim3(kThreads), lds, stream, bottom_data,
                     spatial_scale, num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois,
                     group_size, output_dim, scratch, nullptr);
  dtt_prof_end("psroi_fwd_plane", stream);
  DTT_CHECK_LAUNCH("psroi forward (channel-major)");
  const long n_out = (long)num_rois * output_dim;
  hipLaunchKernelGGL(psroi_vote_tr, dim3(dtt_cdiv(n_out, 256)), dim3(256), 0, stream, scratch, num_rois, output_dim,
                     group_size, pooled_height, pooled_width, vote_out);
  DTT_CHECK_LAUNCH("psroi vote (channel-major)");
  return 1;
}

extern "C" int dtt_psroi_pool_backward(const float* top_diff, const int* mapping_channel, int batch_size,
                                       int num_rois, float spatial_scale, int channels, int height, int width,
                                       int pooled_width, int pooled_height, int output_dim, int group_size,
                                       float* bottom_diff, const float* bottom_rois, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!check_common(batch_size, num_rois, height, width, channels, pooled_height, pooled_width, group_size, output_dim))
    return 0;
  DTT_REQUIRE(bottom_diff, "psroi backward: null bottom_diff");
  DTT_REQUIRE(num_rois == 0 || (top_diff && bottom_rois), "psroi backward: null pointer");
  if (plane_path_ok(height, width, pooled_height, pooled_width, group_size)) {
    const size_t lds = (size_t)height * width * sizeof(float);
    if (!raise_lds(reinterpret_cast<const void*>(psroi_bwd_plane), lds)) return 0;
    hipLaunchKernelGGL(psroi_bwd_plane, dim3(channels, batch_size), dim3(kThreads), lds, stream, top_diff,
                       spatial_scale, num_rois, height, width, channels, pooled_height, pooled_width, bottom_rois,
                       group_size, output_dim, bottom_diff);
  } else {
    hipError_t e = hipMemsetAsync(bottom_diff, 0, (size_t)batch_size * channels * height * width * sizeof(float), stream);
    DTT_REQUIRE(e == hipSuccess, "psroi backward: memset failed: %s", hipGetErrorString(e));
    const long n = (long)num_rois * output_dim * pooled_height * pooled_width;
    if (n > 0)
      hipLaunchKernelGGL(psroi_bwd_generic, dim3(min(dtt_cdiv(n, 256), 65535)), dim3(256), 0, stream, n, top_diff,
                         mapping_channel, spatial_scale, height, width, channels, pooled_height, pooled_width,
                         group_size, output_dim, bottom_diff, bottom_rois);
  }
  DTT_CHECK_LAUNCH("psroi backward");
  return 1;
}

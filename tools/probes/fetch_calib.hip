// FETCH_SIZE calibration for the channels-last correlation's access pattern (MI355X_MICROARCH.md, HBM section: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   stream64 : every wave reads 16 pixels x 64 B (4 lanes x 16 B per pixel), pixels 8 KB apart -- one 16-channel chunk of a
//              2048-channel channels-last map, the DMA pattern of corr_nhwc_kernel; 64 B requested per 128-B line
//   stream128: the same pixels, both 64-B halves of the line (two loads)
//   linear   : a fully coalesced 16 B/lane streaming read of the same number of requested bytes as stream64
//   stream256: 16 lanes x 16 B = 256 B per pixel (one 64-channel group: two whole 128-B lines), pixels 8 KB apart, plain loads
//   stream256_lds: the same bytes by global_load_lds_dwordx4 (the halo stream of corr_bwd_stream_kernel, csrc/correlation_bwd.hip)
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/probes/fetch_calib.hip ; run under
// rocprofv3 --kernel-trace --pmc FETCH_SIZE.  Prints the requested bytes of each kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void stream64(const float* __restrict__ p, long npix, long pix_stride, float* sink) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long pix = w * 16 + (lane >> 2);
  f32x4 v = {0, 0, 0, 0};
  if (pix < npix) v = *reinterpret_cast<const f32x4*>(p + pix * pix_stride + (lane & 3) * 4);
  if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = 1.f;
}
__global__ void stream128(const float* __restrict__ p, long npix, long pix_stride, float* sink) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long pix = w * 16 + (lane >> 2);
  f32x4 v = {0, 0, 0, 0}, u = {0, 0, 0, 0};
  if (pix < npix) {
    v = *reinterpret_cast<const f32x4*>(p + pix * pix_stride + (lane & 3) * 4);
    u = *reinterpret_cast<const f32x4*>(p + pix * pix_stride + 16 + (lane & 3) * 4);
  }
  if (v.x + v.y + v.z + v.w + u.x + u.y + u.z + u.w == 12345.f) sink[0] = 1.f;
}
__global__ void linear(const float* __restrict__ p, long n4, float* sink) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 v = {0, 0, 0, 0};
  if (i < n4) v = reinterpret_cast<const f32x4*>(p)[i];
  if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = 1.f;
}

__global__ void stream256(const float* __restrict__ p, long npix, long pix_stride, float* sink) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long pix = w * 4 + (lane >> 4);
  f32x4 v = {0, 0, 0, 0};
  if (pix < npix) v = *reinterpret_cast<const f32x4*>(p + pix * pix_stride + (lane & 15) * 4);
  if (v.x + v.y + v.z + v.w == 12345.f) sink[0] = 1.f;
}
__global__ void stream256_lds(const float* __restrict__ p, long npix, long pix_stride, float* sink) {
  __shared__ float buf[4 * 256];   // 1 KB per wave
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + wv;
  const long pix = w * 4 + (lane >> 4);
  const unsigned voff = (unsigned)((lane >> 4) * pix_stride * 4 + (lane & 15) * 16);
  const char* base = reinterpret_cast<const char*>(p + (w * 4) * pix_stride);
  const unsigned long long bv = (unsigned long long)base;
  const char* sbase = (const char*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)bv));
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(buf + wv * 256));
  if (pix < npix)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_waitcnt vmcnt(0)" : : "s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
  __syncthreads();
  if (buf[threadIdx.x] == 12345.f) sink[0] = 1.f;
}

int main() {
  const long npix = 1L << 20, stride = 2048;                 // floats: 8 KB per pixel -> an 8 GB map
  float *buf, *sink;
  if (hipMalloc(&buf, npix * stride * sizeof(float)) != hipSuccess) return 1;
  hipMalloc(&sink, 64);
  hipMemset(buf, 0, npix * stride * sizeof(float));
  hipDeviceSynchronize();
  const int wg = 256, wpb = wg / 64;
  const long waves = npix / 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(stream64, dim3(waves / wpb), dim3(wg), 0, 0, buf, npix, stride, sink);
    hipLaunchKernelGGL(stream128, dim3(waves / wpb), dim3(wg), 0, 0, buf, npix, stride, sink);
    hipLaunchKernelGGL(stream256, dim3(npix / 4 / wpb), dim3(wg), 0, 0, buf, npix, stride, sink);
    hipLaunchKernelGGL(stream256_lds, dim3(npix / 4 / wpb), dim3(wg), 0, 0, buf, npix, stride, sink);
    hipLaunchKernelGGL(linear, dim3(npix * 16 / 4 / wg), dim3(wg), 0, 0, buf + (long)(rep + 1) * (1L << 28), npix * 16 / 4, sink);
  }
  hipDeviceSynchronize();
  printf("requested bytes: stream64 %ld  stream128 %ld  stream256 %ld  stream256_lds %ld  linear %ld\n", npix * 64, npix * 128, npix * 256,
         npix * 256, npix * 64);
  return 0;
}

// WRITE_SIZE calibration for the store patterns of the hand-written kernels (MI355X_MICROARCH.md, HBM section: "WRITE_SIZE is
// uncalibrated: calibrate on a known byte count in your own access pattern").
//   half64  : every wave stores 16 rows x 64 B (lane = (row, 16-byte piece)), rows 7168 B apart, the other half of each
//             128-B line never written -- one accumulator tile of head_gemm_kernel's epilogue
//   pair128 : the same wave stores both 64-B halves of the line back to back (two instructions)
//   linear  : a fully coalesced 16 B/lane streaming store (16 x 16 B per row: four times the bytes of half64)
//   runs17  : every wave stores 64 consecutive floats starting at an arbitrary 4-byte offset of a 1184-float row -- the
//             window-split correlation's write-out (partial lines at both ends of a run)
// Build: hipcc --offload-arch=gfx950 -O3 -o write_calib tools/probes/write_calib.hip ; run under
// rocprofv3 --kernel-trace --pmc WRITE_SIZE.  Prints the stored bytes of each kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void half64(float* p, long nrows, long row_stride) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long row = w * 16 + (lane & 15);
  if (row < nrows) *reinterpret_cast<f32x4*>(p + row * row_stride + (lane >> 4) * 4) = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ void pair128(float* p, long nrows, long row_stride) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const long row = w * 16 + (lane & 15);
  if (row < nrows) {
    *reinterpret_cast<f32x4*>(p + row * row_stride + (lane >> 4) * 4) = f32x4{1.f, 2.f, 3.f, 4.f};
    *reinterpret_cast<f32x4*>(p + row * row_stride + 16 + (lane >> 4) * 4) = f32x4{1.f, 2.f, 3.f, 4.f};
  }
}
__global__ void linear(float* p, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
__global__ void runs17(float* p, long nruns, long row_stride) {
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w < nruns) p[w * row_stride + 473 + (w % 3) * 85 + lane] = 1.f;
}

int main() {
  const long nrows = 1L << 20, stride = 1792;               // floats: 7 KB per row -> 7.5 GB
  float* buf;
  if (hipMalloc(&buf, nrows * stride * sizeof(float)) != hipSuccess) return 1;
  hipMemset(buf, 0, nrows * stride * sizeof(float));
  hipDeviceSynchronize();
  const int wg = 256, wpb = wg / 64;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(half64, dim3(nrows / 16 / wpb), dim3(wg), 0, 0, buf, nrows, stride);
    hipLaunchKernelGGL(pair128, dim3(nrows / 16 / wpb), dim3(wg), 0, 0, buf + 64, nrows, stride);
    hipLaunchKernelGGL(linear, dim3(nrows * 16 / wg), dim3(wg), 0, 0, buf + (long)(rep + 1) * (1L << 28), nrows * 16);
    hipLaunchKernelGGL(runs17, dim3(nrows / wpb), dim3(wg), 0, 0, buf, nrows, 1184L);
  }
  hipDeviceSynchronize();
  printf("stored bytes: half64 %ld  pair128 %ld  linear %ld  runs17 %ld\n", nrows * 64, nrows * 128, nrows * 256, nrows * 256);
  return 0;
}

// Probe: what one LDS read costs a single wave per SIMD that is otherwise issuing v_mfma_f32_16x16x4_f32 back to back
// (developer tool).  NR reads of width WB bytes per block of 20 MFMAs; results are consumed only at the end.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NR, int WB>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 15) * 0.001f;
  __syncthreads();
  f32x4 acc[40];
#pragma unroll
  for (int i = 0; i < 40; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 bx[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) bx[q] = *reinterpret_cast<f32x4*>(&lds[(lane * 4 + q * 256) & 16383]);
  f32x4 a = *reinterpret_cast<f32x4*>(&lds[lane * 4 + 8192]);
  f32x4 sink = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      f32x4 r[NR > 0 ? NR : 1];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int off = (lane * 4 + blk * 512 + n * 1024 + it * 64) & 16380;
        if (WB == 16) r[n] = *reinterpret_cast<f32x4*>(&lds[off]);
        else if (WB == 8) { f32x2 t = *reinterpret_cast<f32x2*>(&lds[off]); r[n] = f32x4{t[0], t[1], 0, 0}; }
        else r[n] = f32x4{lds[off], 0, 0, 0};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 5; ++q)
          acc[blk * 5 + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bx[q][j], acc[blk * 5 + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NR; ++n) sink += r[n];   // VALU after the block (4 v_add per read)
    }
  }
  f32x4 s = sink;
#pragma unroll
  for (int i = 0; i < 40; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NR, int WB>
void run() {
  float* out;
  (void)hipMalloc(&out, 256 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR, WB>), dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)iters * 160;
  printf("%d reads of %2d B per 20 MFMAs: %.2f ns/MFMA = %.1f cycles @2.4GHz  (block of 20: %.0f cycles)\n", NR, WB, ms * 1e6 / mfma,
         ms * 1e6 / mfma * 2.4, ms * 1e6 / mfma * 2.4 * 20);
  (void)hipFree(out);
}

int main() {
  run<0, 16>(); run<1, 16>(); run<2, 16>(); run<4, 16>(); run<1, 8>(); run<2, 8>(); run<4, 8>(); run<1, 4>(); run<4, 4>();
  return 0;
}

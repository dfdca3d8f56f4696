// Probe: what one s_barrier costs a workgroup of NW waves (a) alone, (b) between blocks of back-to-back
// v_mfma_f32_16x16x4_f32 issued by CW of its waves (developer tool; prints cycles per loop iteration from s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NMFMA>
__global__ void k(unsigned long long* out, float* sink, int iters, int cw) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const float a = (float)threadIdx.x, b = 1.0f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_barrier();
    if (wave < cw) {
#pragma unroll
      for (int m = 0; m < NMFMA; ++m) acc[m % 9] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 9], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) s += acc[i];
  if (s[0] == 12345.f) sink[threadIdx.x] = s[1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NMFMA>
void run(int nw, int cw) {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NMFMA>), dim3(256), dim3(nw * 64), 0, 0, out, sink, iters, cw);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[256]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%2d waves, %d compute waves x %2d MFMAs per barrier: %7.1f s_memtime ticks / iteration, %7.1f ns / iteration (MFMA floor %d cycles per SIMD)\n",
         nw, cw, NMFMA, (double)h[0] / iters, ms * 1e6 / iters, NMFMA * 32 * ((cw + 3) / 4));
  (void)hipFree(out); (void)hipFree(sink);
}

int main() {
  run<0>(4, 0); run<0>(8, 0); run<0>(12, 0); run<0>(16, 0);
  run<36>(4, 4); run<36>(8, 4); run<36>(12, 4);
  run<18>(8, 8); run<18>(12, 8); run<36>(12, 8); run<72>(12, 8); run<9>(12, 12); run<18>(12, 12);
  return 0;
}

// Probe 2: the window-split correlation's compute interval in isolation -- 8 compute waves (2 per SIMD) + 4 idle waves,
// one s_barrier per interval; in every interval one wave of each SIMD pair issues NREAD ds_read_b128 (consumed by its
// MFMAs one interval later) + 18 MFMAs, the other 18 MFMAs.  Prints shader cycles per interval (MFMA floor 1152).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NREAD, bool DEP>
__global__ __launch_bounds__(768) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[32768];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  f32x4 acc[9], X[9], Y[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { acc[i] = f32x4{0, 0, 0, 0}; X[i] = f32x4{1.f, 2.f, 3.f, (float)lane}; Y[i] = X[i]; }
  const int kh = wave >> 2;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 8) {
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ph == kh) {
#pragma unroll
          for (int r = 0; r < NREAD; ++r) Y[r] = *reinterpret_cast<const f32x4*>(&lds[((it * 64 + r * 1024 + wave * 4096) & 32767 & ~255) + lane * 4]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int m = 0; m < 9; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[m][s + 2 * (ph != kh)], X[(m + 1) % 9][s], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (DEP && ph != kh) {
#pragma unroll
          for (int r = 0; r < 9; ++r) X[r] = Y[r];
        }
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) s += acc[i] + Y[i];
  if (s[0] == 12345.f) sink[threadIdx.x] = s[1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NREAD, bool DEP>
void run() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 8192);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NREAD, DEP>), dim3(256), dim3(768), 0, 0, out, sink, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long h[256]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("%2d reads per own interval, dep %d: %7.1f shader cycles / interval, %7.1f ns / interval -> %.2f GHz\n", NREAD, (int)DEP,
         (double)h[0] / iters, ms * 1e6 / iters, (double)h[0] / iters / (ms * 1e6 / iters));
  (void)hipFree(out); (void)hipFree(sink);
}

int main() {
  run<0, false>(); run<1, false>(); run<5, false>(); run<9, false>(); run<9, true>();
  return 0;
}

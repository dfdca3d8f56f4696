// Probe 3: as probe 2 (8 compute waves, 2 per SIMD, one barrier per interval, 18 MFMAs per wave per interval), sweeping
// how the reading wave's NREAD LDS reads are issued: WB bytes per lane, MODE 0 = one burst before the MFMAs,
// 1 = one read after every second MFMA, 2 = burst AFTER the MFMAs, 3 = both waves read NREAD/2 every interval (burst first).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NREAD, int WB, int MODE, int PATTERN = 0>
__global__ __launch_bounds__(768) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[32768];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  f32x4 acc[9], X[9], Y[12];
#pragma unroll
  for (int i = 0; i < 9; ++i) { acc[i] = f32x4{0, 0, 0, 0}; X[i] = f32x4{1.f, 2.f, 3.f, (float)lane}; }
#pragma unroll
  for (int i = 0; i < 12; ++i) Y[i] = f32x4{1.f, 2.f, 3.f, (float)lane};
  const int kh = wave >> 2;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  auto rd = [&](int r, int it) {
    const float* p;
    if (MODE == 3 && PATTERN == 1) {   // the correlation kernel's pattern: a 4 x 4 pixel block of a 24-pixel-wide halo, 64 B per pixel, swizzled pieces
      const int li = lane & 15, lg = lane >> 4, iy = li >> 2, ix = li & 3;
      const int sw = (0x1320 >> ((iy & 3) * 4)) & 3;
      const int row0 = 4 * ((r + (wave >> 1)) % 3), col0 = 4 * ((r * 2 + wave) % 5);
      p = &lds[(((row0 + iy) * 24 + col0 + ix) * 16 + ((lg ^ sw) << 2) + ((it >> 1) & 3) * 7168) & 32767];
    } else if (MODE == 3 && PATTERN == 2) {   // same blocks, no swizzle
      const int li = lane & 15, lg = lane >> 4, iy = li >> 2, ix = li & 3;
      const int row0 = 4 * ((r + (wave >> 1)) % 3), col0 = 4 * ((r * 2 + wave) % 5);
      p = &lds[(((row0 + iy) * 24 + col0 + ix) * 16 + (lg << 2) + ((it >> 1) & 3) * 7168) & 32767];
    } else p = &lds[((it * 64 + r * 1024 + wave * 4096) & 32767 & ~255) + lane * (WB / 4)];
    if (WB == 16) Y[r] = *reinterpret_cast<const f32x4*>(p);
    else { f32x2 t = *reinterpret_cast<const f32x2*>(p); Y[r][0] = t[0]; Y[r][1] = t[1]; }
  };
  if (wave < 8) {
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool reader = MODE == 3 ? true : ph == kh;
        constexpr int NR = MODE == 3 ? (NREAD + 1) / 2 : NREAD;
        if (reader && (MODE == 0 || MODE == 3)) {
#pragma unroll
          for (int r = 0; r < NR; ++r) rd(r, it);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int m = 0; m < 9; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[m][s + 2 * (ph != kh)], X[(m + 1) % 9][s], acc[m], 0, 0, 0);
            if (MODE == 1 && reader && ((s * 9 + m) & 1) == 1 && (s * 9 + m) / 2 < NREAD) { __builtin_amdgcn_sched_barrier(0); rd((s * 9 + m) / 2, it); __builtin_amdgcn_sched_barrier(0); }
          }
        __builtin_amdgcn_sched_barrier(0);
        if (reader && MODE == 2) {
#pragma unroll
          for (int r = 0; r < NR; ++r) rd(r, it);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) s += acc[i] + Y[i];
  s += Y[9] + Y[10] + Y[11];
  if (s[0] == 12345.f) sink[threadIdx.x] = s[1];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NREAD, int WB, int MODE, int PATTERN = 0>
void run() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 8192);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<NREAD, WB, MODE, PATTERN>), dim3(256), dim3(768), 0, 0, out, sink, iters);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[256]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("reads %2d x %2d B, mode %d, pattern %d: %7.1f cycles / interval\n", NREAD, WB, MODE, PATTERN, (double)h[0] / iters);
  (void)hipFree(out); (void)hipFree(sink);
}

int main() {
  run<10, 16, 3, 0>(); run<10, 16, 3, 1>(); run<10, 16, 3, 2>(); run<4, 16, 3, 1>(); run<2, 16, 3, 1>();
  return 0;
}

// Probe: sustained issue rate of v_mfma_f32_16x16x4_f32 for the head GEMM's accumulator pattern (developer tool).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Variants: NACC accumulators round-robin (operands in registers), with / without an LDS operand read per block,
// 4 waves per workgroup (one per SIMD) or 6 (two parked at a barrier like the loader waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int TPG, bool LDSREAD, bool BARRIER, bool RANDOM>
__global__ __launch_bounds__(384) void k(float* out, int iters, int nbar) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
    unsigned hsh = (unsigned)(i + blockIdx.x * 16384) * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    lds[i] = RANDOM ? ((float)(hsh & 0xffffff) / 8388608.f - 1.f) : (float)(i & 15) * 0.001f;
  }
  __syncthreads();
  if (wave >= 4) {
    for (int s = 0; s < nbar; ++s) __builtin_amdgcn_s_barrier();
    return;
  }
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 bx[TPG];
#pragma unroll
  for (int q = 0; q < TPG; ++q) bx[q] = *reinterpret_cast<f32x4*>(&lds[(lane * 4 + q * 256) & 16383]);
  f32x4 a = *reinterpret_cast<f32x4*>(&lds[lane * 4 + 8192]);
  for (int it = 0; it < iters; ++it) {
    if (BARRIER) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int blk = 0; blk < NACC / TPG; ++blk) {
      f32x4 an = a;
      if (LDSREAD) an = *reinterpret_cast<f32x4*>(&lds[(lane * 4 + blk * 512 + it * 64) & 16383]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < TPG; ++q)
          acc[blk * TPG + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bx[q][j], acc[blk * TPG + q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a = an;
    }
  }
  f32x4 s = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, int TPG, bool LDSREAD, bool BARRIER, bool RANDOM = false>
void run(const char* name, int threads) {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, TPG, LDSREAD, BARRIER, RANDOM>), dim3(256), dim3(threads), 0, 0, out, iters, BARRIER ? iters : 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)iters * NACC * 4;
  printf("%-48s %8.1f us  %.2f ns/MFMA = %.1f cycles @2.4GHz   %.1f TFLOP/s\n", name, ms * 1e3, ms * 1e6 / mfma, ms * 1e6 / mfma * 2.4,
         mfma * 1024 * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<40, 5, false, false>("40 acc, groups of 5, regs only, 4 waves", 256);
  run<40, 5, true, false>("40 acc, groups of 5, LDS a-read, 4 waves", 256);
  run<40, 5, true, true>("40 acc, groups of 5, LDS read + barrier, 4 waves", 256);
  run<40, 5, true, true>("40 acc, groups of 5, LDS read + barrier, 6 waves", 384);
  run<40, 5, false, false>("40 acc, groups of 5, regs only, 6 waves", 384);
  run<40, 5, true, true, true>("40 acc, groups of 5, LDS read + barrier, 6 waves, RANDOM data", 384);
  run<40, 5, false, false, true>("40 acc, groups of 5, regs only, 4 waves, RANDOM data", 256);
  run<20, 5, false, false>("20 acc, groups of 5, regs only, 4 waves", 256);
  run<40, 10, false, false>("40 acc, groups of 10, regs only, 4 waves", 256);
  run<8, 4, false, false>("8 acc, groups of 4, regs only, 4 waves", 256);
  run<4, 4, false, false>("4 acc, groups of 4 (dependent every 4), 4 waves", 256);
  return 0;
}

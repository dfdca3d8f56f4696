// Probe: where does the workgroup dispatcher put the workgroups of a small kernel (4 x 1024 threads, 64 KB LDS: the
// proposal layer's select / sort) launched beside a one-workgroup-per-CU kernel (G x 768 threads, 147 KB LDS: the
// correlation / head GEMM shape), and does it wait for a CU although some are empty?
//   hipcc --offload-arch=gfx950 -O3 -o wg_placement tools/probes/wg_placement.hip && ./wg_placement
// Every workgroup records (XCC, SE, CU) from HW_ID, its start time and its end time (s_memrealtime, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Rec { unsigned xcc, se, cu; unsigned long long t0, t1; };

__device__ __forceinline__ void hwid(Rec& r) {
  unsigned id, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  r.cu = (id >> 8) & 15; r.se = (id >> 13) & 7; r.xcc = xcc & 15;
}

__global__ void spin(Rec* out, unsigned long long ticks) {
  extern __shared__ float lds[];
  Rec r;
  r.t0 = __builtin_readcyclecounter();
  unsigned long long s = wall_clock64();
  r.t0 = s;
  hwid(r);
  lds[threadIdx.x] = 1.f;
  while (wall_clock64() - s < ticks) __builtin_amdgcn_s_sleep(8);
  r.t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  Rec *da, *db;
  hipMalloc(&da, 4096 * sizeof(Rec)); hipMalloc(&db, 64 * sizeof(Rec));
  const unsigned long long us = 100;  // wall_clock64 ticks per microsecond (100 MHz)
  for (int order = 0; order < 2; ++order)
    for (int G : {256, 248, 240, 224, 192}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        if (order == 0) {
          hipLaunchKernelGGL(spin, dim3(G), dim3(768), 147 * 1024, a, da, 100 * us);
          hipLaunchKernelGGL(spin, dim3(4), dim3(1024), 64 * 1024, b, db, 50 * us);
        } else {
          hipLaunchKernelGGL(spin, dim3(4), dim3(1024), 64 * 1024, b, db, 50 * us);
          hipLaunchKernelGGL(spin, dim3(G), dim3(768), 147 * 1024, a, da, 100 * us);
        }
        hipDeviceSynchronize();
      }
      std::vector<Rec> ra(G), rb(4);
      hipMemcpy(ra.data(), da, G * sizeof(Rec), hipMemcpyDeviceToHost);
      hipMemcpy(rb.data(), db, 4 * sizeof(Rec), hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull;
      for (auto& r : ra) t0 = std::min(t0, r.t0);
      for (auto& r : rb) t0 = std::min(t0, r.t0);
      int late = 0; double latest = 0, end = 0;
      int per[8][8] = {};
      for (auto& r : ra) {
        double st = (r.t0 - t0) / 100.0;
        if (st > 20) ++late;
        latest = std::max(latest, st); end = std::max(end, (r.t1 - t0) / 100.0);
        per[r.xcc & 7][r.se & 7]++;
      }
      printf("%s G=%3d: big kernel: %d workgroups started > 20 us late (latest start %.1f us, end %.1f us)\n",
             order ? "small first" : "big first  ", G, late, latest, end);
      for (auto& r : rb)
        printf("      small wg on xcc %u se %u cu %2u: start %.1f us end %.1f us   (big wgs on that xcc/se: %d)\n", r.xcc, r.se, r.cu,
               (r.t0 - t0) / 100.0, (r.t1 - t0) / 100.0, per[r.xcc & 7][r.se & 7]);
      if (G == 240 && order == 0) {
        printf("      big wgs per xcc x se:");
        for (int x = 0; x < 8; ++x) { printf(" ["); for (int s = 0; s < 4; ++s) printf("%d ", per[x][s]); printf("]"); }
        printf("\n");
      }
    }
  return 0;
}

// Small-problem NMS pieces shared by the workgroup-per-problem kernels (postprocess.hip, tubes.hip): devIoU exactly as
// nms_cuda_kernel.cu:31-39 (FP contraction off), 64-bit IoU mask rows in LDS, and the register-resident greedy sweep.
#pragma once
#include "common.h"

namespace dtt_small_nms {

__device__ __forceinline__ float dev_iou(const float4 a, const float4 b) {
  float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a.z - a.x + 1) * (a.w - a.y + 1);
  float Sb = (b.z - b.x + 1) * (b.w - b.y + 1);
  return interS / (Sa + Sb - interS);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Greedy NMS over box[0..n) (priority = index order) entirely in LDS.  mask: n rows x W words; kept: indices of the
// survivors in order; ctl[1] receives their number.  All threads of the workgroup must call; ends with a barrier.
__device__ __forceinline__ void lds_mask_and_sweep(const float4* box, int n, int W, float nms_thresh,
                                                   unsigned long long* mask, unsigned short* kept, int* ctl,
                                                   int tid, int nthreads) {
  const int lane = tid & 63;
  // ---- IoU mask rows: row i, word w: bit b set iff IoU(i, 64w+b) > thresh and 64w+b > i
  const int nw = (n + 63) >> 6;
  for (int idx = tid; idx < n * nw; idx += nthreads) {
    const int i = idx / nw, w = idx - i * nw;
    unsigned long long bits = 0;
    if (w >= (i >> 6)) {
      const float4 a = box[i];
      const int c0 = w << 6, c1 = min(n, c0 + 64);
      for (int c = max(c0, i + 1); c < c1; ++c)
        if (dev_iou(a, box[c]) > nms_thresh) bits |= 1ULL << (c - c0);
    }
    mask[(size_t)i * W + w] = bits;
  }
  __syncthreads();
  // ---- greedy sweep by wave 0 (nms_cuda_kernel.cu:131-144); lanes < nw hold the removal words
  if (tid < 64) {
    unsigned long long Rw = 0;
    int nk = 0;
    for (int c = 0; c < nw; ++c) {
      const unsigned long long r = readlane64(Rw, c);
      const int rows_c = min(64, n - c * 64);
      const unsigned long long valid = rows_c == 64 ? ~0ULL : ((1ULL << rows_c) - 1ULL);
      const unsigned long long d = lane < rows_c ? mask[(size_t)(c * 64 + lane) * W + c] : 0ULL;
      const unsigned long long nz = __ballot(d != 0ULL);
      unsigned long long alive = ~r & valid;
      unsigned long long cand = alive & nz;
      while (cand != 0) {
        const int i = __builtin_ctzll(cand);
        alive &= ~readlane64(d, i);
        const unsigned long long above = (i == 63) ? 0ULL : (~0ULL << (i + 1));
        cand = alive & nz & above;
      }
      if ((alive >> lane) & 1ULL) kept[nk + __builtin_popcountll(alive & ((1ULL << lane) - 1ULL))] = (unsigned short)(c * 64 + lane);
      nk += __builtin_popcountll(alive);
      if (c + 1 < nw) {
        const bool owner = lane > c && lane < nw;
        unsigned long long kk = alive, accw = 0;
        while (kk != 0) {
          const int i = __builtin_ctzll(kk);
          kk &= kk - 1;
          accw |= mask[(size_t)(c * 64 + i) * W + (owner ? lane : 0)];
        }
        if (owner) Rw |= accw;
      }
    }
    if (lane == 0) ctl[1] = nk;
  }
  __syncthreads();
}

}  // namespace dtt_small_nms

// Developer probe: does global_load_lds_dwordx4 accept 4-byte-aligned global sources on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ void probe(const float* src, float* out, int shift, int stride) {
  __shared__ __attribute__((aligned(16))) float buf[256 + 8];
  const int lane = threadIdx.x;
  const float* p = src + shift + lane * stride;   // 16-byte piece at 4-byte alignment
  __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)buf, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);   // everything
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = buf[lane * 4 + j];
}
int main() {
  const int N = 4096;
  std::vector<float> h(N);
  for (int i = 0; i < N; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int stride : {4, 5, 7, 20}) for (int shift : {0, 1, 2, 3, 67}) {
    hipMemset(o, 0, 256 * 4);
    probe<<<1, 64>>>(d, o, shift, stride);
    std::vector<float> r(256);
    hipError_t e = hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (r[l * 4 + j] != (float)(shift + l * stride + j)) ++bad;
    printf("stride %2d shift %2d: %s (%d bad) err=%d  first: %.0f %.0f %.0f %.0f\n", stride, shift, bad ? "MISMATCH" : "ok", bad, (int)e, r[0], r[1], r[2], r[3]);
    bad_total += bad;
  }
  return bad_total != 0;
}

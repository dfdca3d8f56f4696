// The two RPN losses of rpn/rpn.py:86-105 and their gradient, hand-written (gfx950): the class loss = cross-entropy over the anchors
// the anchor-target layer sampled (label 0 / 1; -1 = not sampled; rpn.py:90-97 gathers them with nonzero() + index_select, the mean over
// the labelled anchors is the same number) and the box loss = _smooth_l1_loss(..., sigma, dim = [1, 2, 3]) (net_utils.py:73-87,
// rpn.py:104-105), for several LEGS (frames of a pair) in one launch -- each leg has its own two scalars.
//
//   forward   one pass over cls_prob / labels / bbox_pred / targets / weights: per-workgroup partial sums in a fixed layout, the last
//             workgroup to finish (a ticket) adds them in index order: deterministic, no float atomics.
//             class loss per anchor = -log p[label], p = the pairwise softmax the head GEMM's epilogue applied (heads.hip).
//   backward  the gradient with respect to the score LOGITS, (p - y) * g / count -- not the gradient with respect to p pushed through
//             the softmax's adjoint, which multiplies by p and dies where p underflows; the reference's cross_entropy on the logits
//             keeps -g / count there -- and the smooth-L1 gradient with respect to bbox_pred, both as dense NCHW maps for
//             dtt_rpn_head_grad_rows (cls_grad_is_logits = 1).
#include <float.h>
#include "common.h"

namespace {

constexpr int kLossThreads = 256;
constexpr int kLossZ = 4;   // anchor slices per pixel block (grid z)

struct LossGeom {
  const float *prob, *bbox, *labels, *tgt, *w_in, *w_out;
  int batch, legs, A, hw;
  float s2;        // sigma^2
};

__device__ __forceinline__ float block_sum(float v, float* sh) {   // every thread gets the total; fixed order
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < kLossThreads / 64; ++i) t += sh[i];
  return t;
}

__global__ __launch_bounds__(kLossThreads) void rpn_loss_fwd_kernel(LossGeom g, float* __restrict__ partial, unsigned* __restrict__ ticket,
                                                                    float* __restrict__ loss, float* __restrict__ count) {
  __shared__ float sh[kLossThreads / 64];
  __shared__ int last;
  const int p = blockIdx.x * kLossThreads + threadIdx.x, b = blockIdx.y, z = blockIdx.z;
  float cls = 0.f, cnt = 0.f, box = 0.f;
  if (p < g.hw) {
    for (int a = z; a < g.A; a += kLossZ) {
      const float lab = g.labels[((long)b * g.A + a) * g.hw + p];
      if (lab >= 0.f) {
        const float pl = g.prob[((long)b * 2 * g.A + (lab > 0.5f ? g.A : 0) + a) * g.hw + p];
        cls -= logf(fmaxf(pl, FLT_MIN));
        cnt += 1.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long o = ((long)b * 4 * g.A + 4 * a + c) * g.hw + p;
        const float d = g.w_in[o] * (g.bbox[o] - g.tgt[o]), ad = fabsf(d);
        box += g.w_out[o] * (ad < 1.0f / g.s2 ? d * d * (g.s2 * 0.5f) : ad - 0.5f / g.s2);
      }
    }
  }
  cls = block_sum(cls, sh); cnt = block_sum(cnt, sh); box = block_sum(box, sh);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    float* o = partial + (((long)b * kLossZ + z) * nblk + blockIdx.x) * 4;
    o[0] = cls; o[1] = cnt; o[2] = box;
    __threadfence();
    const unsigned t = atomicAdd(ticket, 1u);
    last = t == gridDim.x * gridDim.y * gridDim.z - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last workgroup: leg by leg, every thread adds its fixed share of the leg's partials (index t, t + 256, ...) in double, then a
  // fixed-shape tree over the 256 shares -- the same order of additions whichever workgroup comes last.  (The partials are read with
  // agent-scope loads: past this CU's L1, nothing serialised -- a volatile walk by one lane waited for every load in turn.)
  __shared__ double tree[3][kLossThreads];
  const int per = g.batch / g.legs;
  const long per_leg = (long)per * kLossZ * nblk;
  for (int leg = 0; leg < g.legs; ++leg) {
    double sc = 0.0, sn = 0.0, sb = 0.0;
    for (long i = (long)leg * per_leg + threadIdx.x; i < (long)(leg + 1) * per_leg; i += kLossThreads) {
      sc += (double)__hip_atomic_load(partial + i * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sn += (double)__hip_atomic_load(partial + i * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sb += (double)__hip_atomic_load(partial + i * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    tree[0][threadIdx.x] = sc; tree[1][threadIdx.x] = sn; tree[2][threadIdx.x] = sb;
    __syncthreads();
    for (int o = kLossThreads / 2; o >= 1; o >>= 1) {
      if ((int)threadIdx.x < o) {
        tree[0][threadIdx.x] += tree[0][threadIdx.x + o]; tree[1][threadIdx.x] += tree[1][threadIdx.x + o];
        tree[2][threadIdx.x] += tree[2][threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      loss[leg] = (float)(tree[0][0] / tree[1][0]);             // (no labelled anchor: 0 / 0 = nan, as F.cross_entropy's mean over nothing)
      loss[g.legs + leg] = (float)(tree[2][0] / (double)per);
      count[leg] = (float)tree[1][0];
    }
  }
  if (threadIdx.x == 0) *ticket = 0u;
}

__global__ __launch_bounds__(kLossThreads) void rpn_loss_bwd_kernel(LossGeom g, const float* __restrict__ grad_loss, const float* __restrict__ count,
                                                                    float* __restrict__ g_logits, float* __restrict__ g_bbox) {
  const int p = blockIdx.x * kLossThreads + threadIdx.x, b = blockIdx.y;
  if (p >= g.hw) return;
  const int per = g.batch / g.legs, leg = b / per;
  const float sc = grad_loss[leg] / count[leg], sb = grad_loss[g.legs + leg] / (float)per;
  for (int a = blockIdx.z; a < g.A; a += kLossZ) {
    const float lab = g.labels[((long)b * g.A + a) * g.hw + p];
    const long ob = ((long)b * 2 * g.A + a) * g.hw + p, of = ob + (long)g.A * g.hw;
    float gb = 0.f, gf = 0.f;
    if (lab >= 0.f) {
      const bool fg = lab > 0.5f;
      gb = (g.prob[ob] - (fg ? 0.f : 1.f)) * sc;
      gf = (g.prob[of] - (fg ? 1.f : 0.f)) * sc;
    }
    g_logits[ob] = gb; g_logits[of] = gf;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long o = ((long)b * 4 * g.A + 4 * a + c) * g.hw + p;
      const float wi = g.w_in[o], d = wi * (g.bbox[o] - g.tgt[o]), ad = fabsf(d);
      const float dl = ad < 1.0f / g.s2 ? g.s2 * d : (d > 0.f ? 1.f : -1.f);
      g_bbox[o] = g.w_out[o] * wi * dl * sb;
    }
  }
}

}  // namespace

// bytes of workspace dtt_rpn_loss_forward needs
extern "C" size_t dtt_rpn_loss_workspace_bytes(int batch, int hw) {
  if (batch < 1 || hw < 1) return 0;
  return ((size_t)batch * kLossZ * dtt_cdiv(hw, kLossThreads) * 4 + 4) * sizeof(float);
}

// rpn.py:86-105 for `legs` legs of batch / legs images each (image b belongs to leg b / (batch / legs)).
// cls_prob (batch, 2A, H, W): channel a = background, A + a = foreground of anchor a, pairwise softmaxed; labels (batch, 1, A*H, W)
// floats in {1, 0, -1}; bbox_pred / bbox_targets / inside / outside weights (batch, 4A, H, W).
// loss[leg] = class loss, loss[legs + leg] = box loss; count[leg] = labelled anchors of the leg (the backward's divisor).
extern "C" int dtt_rpn_loss_forward(const float* cls_prob, const float* bbox_pred, const float* labels, const float* bbox_targets,
                                    const float* inside_weights, const float* outside_weights, int batch, int legs, int num_anchors,
                                    int hw, float sigma, float* loss, float* count, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(cls_prob && bbox_pred && labels && bbox_targets && inside_weights && outside_weights && loss && count, "rpn_loss: null pointer");
  DTT_REQUIRE(batch > 0 && legs > 0 && legs <= 64 && batch % legs == 0 && num_anchors > 0 && hw > 0 && sigma > 0.f, "rpn_loss: bad shape (batch %d, legs %d)", batch, legs);
  DTT_REQUIRE(workspace && workspace_bytes >= dtt_rpn_loss_workspace_bytes(batch, hw) && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0,
              "rpn_loss: workspace of %zu bytes needed (dtt_rpn_loss_workspace_bytes)", dtt_rpn_loss_workspace_bytes(batch, hw));
  const int nblk = dtt_cdiv(hw, kLossThreads);
  float* partial = static_cast<float*>(workspace);
  unsigned* ticket = reinterpret_cast<unsigned*>(partial + (size_t)batch * kLossZ * nblk * 4);
  DTT_REQUIRE(hipMemsetAsync(ticket, 0, sizeof(unsigned), stream) == hipSuccess, "rpn_loss: memset failed");
  const LossGeom g{cls_prob, bbox_pred, labels, bbox_targets, inside_weights, outside_weights, batch, legs, num_anchors, hw, sigma * sigma};
  dtt_prof_begin("rpn_loss", stream);
  hipLaunchKernelGGL(rpn_loss_fwd_kernel, dim3(nblk, batch, kLossZ), dim3(kLossThreads), 0, stream, g, partial, ticket, loss, count);
  dtt_prof_end("rpn_loss", stream);
  DTT_CHECK_LAUNCH("rpn_loss_fwd");
  return 1;
}

// Gradient of sum_leg grad_loss[leg] * class loss + grad_loss[legs + leg] * box loss: grad_logits (batch, 2A, H, W) with respect to
// the score LOGITS whose pairwise softmax cls_prob is, grad_bbox (batch, 4A, H, W) with respect to bbox_pred.  Every element written.
extern "C" int dtt_rpn_loss_backward(const float* cls_prob, const float* bbox_pred, const float* labels, const float* bbox_targets,
                                     const float* inside_weights, const float* outside_weights, const float* grad_loss, const float* count,
                                     int batch, int legs, int num_anchors, int hw, float sigma, float* grad_logits, float* grad_bbox,
                                     void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DTT_REQUIRE(cls_prob && bbox_pred && labels && bbox_targets && inside_weights && outside_weights && grad_loss && count && grad_logits && grad_bbox,
              "rpn_loss backward: null pointer");
  DTT_REQUIRE(batch > 0 && legs > 0 && batch % legs == 0 && num_anchors > 0 && hw > 0 && sigma > 0.f, "rpn_loss backward: bad shape");
  const LossGeom g{cls_prob, bbox_pred, labels, bbox_targets, inside_weights, outside_weights, batch, legs, num_anchors, hw, sigma * sigma};
  hipLaunchKernelGGL(rpn_loss_bwd_kernel, dim3(dtt_cdiv(hw, kLossThreads), batch, kLossZ), dim3(kLossThreads), 0, stream, g, grad_loss, count,
                     grad_logits, grad_bbox);
  DTT_CHECK_LAUNCH("rpn_loss_bwd");
  return 1;
}

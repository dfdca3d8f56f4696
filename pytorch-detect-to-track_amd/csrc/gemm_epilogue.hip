// Row-major GEMM with the whole bottleneck epilogue in the GEMM: out = act(a * w + bias[n] (+ residual)).
//
// The 1x1 convolutions of the channels-last trunk are plain library GEMMs over the (pixels, channels) view
// (hipBLASLt; a hand-written kernel would have to beat Tensile's fp32 MFMA tiles to earn its place).  What this
// entry point adds over calling the library through PyTorch is the epilogue the reference's blocks need and
// PyTorch does not expose: frozen-BatchNorm shift + residual add (the GEMM's beta * C operand, C == D allowed)
// + ReLU in one pass (faster_rcnn/resnet.py:100-107 `out = bn3(conv3(out)); out += residual; out = relu(out)`).
// Row-major operands are handed to the column-major library transposed: out^T (n x rows) = w^T (n x k) * a^T
// (k x rows), so the per-channel bias runs along the library's M dimension, where its bias epilogue lives.
#include "common.h"
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

namespace {

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulHeuristicResult_t heur;
  bool ok = false, tuned = false;
};

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;                                     // per device
std::map<std::tuple<int, long, int, int, int, int>, Plan> g_plans;              // (device, rows, k, n, relu, residual)

}  // namespace

namespace {
// tune_scratch != NULL: time the library's candidates once on the caller's operands (products go to tune_scratch, rows * n
// floats; synchronises the stream) and remember the winner for this shape.  NULL: never allocates, never synchronises --
// a shape that was not tuned runs the library's first heuristic.
int gemm_bias_act_impl(float* out, const float* a, const float* w, const float* bias, const float* residual, long rows,
                       int k, int n, int relu, void* workspace, size_t workspace_bytes, hipStream_t stream,
                       float* tune_scratch, bool run);
}

extern "C" int dtt_gemm_bias_act(float* out, const float* a, const float* w, const float* bias,
                                 const float* residual, long rows, int k, int n, int relu, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
  return gemm_bias_act_impl(out, a, w, bias, residual, rows, k, n, relu, workspace, workspace_bytes,
                            static_cast<hipStream_t>(stream_), nullptr, true);
}

// One-time candidate timing for a shape of dtt_gemm_bias_act (call it before the first product of that shape, outside
// any timed or captured region: it launches every candidate a few times and synchronises the stream).  scratch: rows * n
// floats owned by the caller; a / w / bias / residual are only read.
extern "C" int dtt_gemm_tune(const float* a, const float* w, const float* bias, const float* residual, long rows, int k,
                             int n, int relu, float* scratch, void* workspace, size_t workspace_bytes, void* stream_) {
  DTT_REQUIRE(scratch, "gemm_tune: scratch (rows * n floats) is required");
  return gemm_bias_act_impl(scratch, a, w, bias, residual, rows, k, n, relu, workspace, workspace_bytes,
                            static_cast<hipStream_t>(stream_), scratch, false);
}

namespace {
int gemm_bias_act_impl(float* out, const float* a, const float* w, const float* bias, const float* residual, long rows,
                       int k, int n, int relu, void* workspace, size_t workspace_bytes, hipStream_t stream,
                       float* tune_scratch, bool run) {
  DTT_REQUIRE(out && a && w && bias, "gemm_bias_act: null pointer");
  DTT_REQUIRE(rows > 0 && k > 0 && n > 0, "gemm_bias_act: bad shape");
  int dev = 0;
  DTT_REQUIRE(hipGetDevice(&dev) == hipSuccess, "gemm_bias_act: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t& handle = g_handles[dev];
  if (!handle) DTT_REQUIRE(hipblasLtCreate(&handle) == HIPBLAS_STATUS_SUCCESS, "gemm_bias_act: hipblasLtCreate failed");
  Plan& p = g_plans[std::make_tuple(dev, rows, k, n, relu, residual ? 1 : 0)];
  if (p.ok && tune_scratch && !p.tuned) {   // built from the first heuristic earlier: redo the selection with timing
    hipblasLtMatmulDescDestroy(p.desc);
    hipblasLtMatrixLayoutDestroy(p.la); hipblasLtMatrixLayoutDestroy(p.lb); hipblasLtMatrixLayoutDestroy(p.lc);
    p = Plan();
  }
  if (!p.ok) {
    DTT_REQUIRE(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS,
                "gemm_bias_act: desc");
    const hipblasOperation_t op = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op, sizeof(op));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op, sizeof(op));
    const hipblasLtEpilogue_t epi = relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    DTT_REQUIRE(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) ==
                    HIPBLAS_STATUS_SUCCESS, "gemm_bias_act: epilogue attribute");
    // column-major views: A = w^T (n x k, ld n), B = a^T (k x rows, ld k), C = D = out^T (n x rows, ld n)
    DTT_REQUIRE(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_32F, n, k, n) == HIPBLAS_STATUS_SUCCESS &&
                    hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_32F, k, rows, k) == HIPBLAS_STATUS_SUCCESS &&
                    hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, n, rows, n) == HIPBLAS_STATUS_SUCCESS,
                "gemm_bias_act: layouts");
    // the bias pointer takes part in kernel selection on some versions: set it before asking
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    hipblasLtMatmulPreference_t pref;
    DTT_REQUIRE(hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS, "gemm_bias_act: preference");
    const uint64_t ws = workspace ? workspace_bytes : 0;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    // Ask for several candidates and time them once on the caller's operands (the library's first heuristic pick
    // is 10-25 % off the best tile for these skinny fp32 shapes); the product goes to a scratch D so the caller's
    // buffers are only read.  DTT_GEMM_AUTOTUNE=0 keeps the first heuristic.
    constexpr int kMaxCand = 48;
    static hipblasLtMatmulHeuristicResult_t cand[kMaxCand];  // under g_mu
    int found = 0;
    const char* tune_env = getenv("DTT_GEMM_AUTOTUNE");
    const bool tune = tune_scratch && !(tune_env && tune_env[0] == '0');
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(handle, p.desc, p.la, p.lb, p.lc, p.lc, pref,
                                                               tune ? kMaxCand : 1, cand, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    DTT_REQUIRE(st == HIPBLAS_STATUS_SUCCESS && found > 0, "gemm_bias_act: no hipBLASLt kernel for %ld x %d x %d", rows,
                k, n);
    int best = 0;
    if (found > 1) {
      float* scratch = tune_scratch;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        const float alpha = 1.f, beta = residual ? 1.f : 0.f;
        const float* c = residual ? residual : scratch;
        float best_ms = 1e30f;
        for (int i = 0; i < found; ++i) {
          if (cand[i].state != HIPBLAS_STATUS_SUCCESS || cand[i].workspaceSize > ws) continue;
          bool good = true;
          constexpr int kReps = 5;
          for (int r = -1; r < kReps && good; ++r) {  // r == -1: untimed first launch (code object load)
            if (r == 0) (void)hipEventRecord(e0, stream);
            good = hipblasLtMatmul(handle, p.desc, &alpha, w, p.la, a, p.lb, &beta, c, p.lc, scratch, p.lc,
                                   &cand[i].algo, workspace, ws, stream) == HIPBLAS_STATUS_SUCCESS;
          }
          float ms = 0.f;
          if (!good || hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
              hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            continue;
          if (ms < best_ms) { best_ms = ms; best = i; }
        }
      }
      (void)hipGetLastError();
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (getenv("DTT_GEMM_AUTOTUNE_VERBOSE"))
        fprintf(stderr, "[dtt] gemm %ld x %d x %d relu=%d res=%d: %d candidates, picked #%d\n", rows, k, n, relu,
                residual ? 1 : 0, found, best);
    }
    p.heur = cand[best];
    p.ok = true;
    p.tuned = tune_scratch != nullptr;
  }
  if (!run) return 1;
  DTT_REQUIRE(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) ==
                  HIPBLAS_STATUS_SUCCESS, "gemm_bias_act: bias pointer");
  DTT_REQUIRE(p.heur.workspaceSize <= (workspace ? workspace_bytes : 0), "gemm_bias_act: workspace too small (%zu < %zu)",
              workspace ? workspace_bytes : (size_t)0, (size_t)p.heur.workspaceSize);
  const float alpha = 1.f, beta = residual ? 1.f : 0.f;
  const float* c = residual ? residual : out;
  const hipblasStatus_t st = hipblasLtMatmul(handle, p.desc, &alpha, w, p.la, a, p.lb, &beta, c, p.lc, out, p.lc,
                                             &p.heur.algo, workspace, workspace ? workspace_bytes : 0, stream);
  DTT_REQUIRE(st == HIPBLAS_STATUS_SUCCESS, "gemm_bias_act: hipblasLtMatmul failed (%d)", (int)st);
  return 1;
}
}  // namespace

// Strided-batched row-major GEMM without epilogue: out[b] (rows, n) = a[b] (rows, k) * w[b] (k, n), operands packed
// back to back (the 16 products of a Winograd F(2x2, 3x3) convolution, csrc/winograd.hip).  Same transposed column-major
// hand-over and the same once-per-shape candidate timing as dtt_gemm_bias_act.
namespace {
std::map<std::tuple<int, int, long, int, int>, Plan> g_batched_plans;   // (device, batch, rows, k, n)
}

namespace {
int gemm_batched_impl(float* out, const float* a, const float* w, int batch, long rows, int k, int n, void* workspace,
                      size_t workspace_bytes, hipStream_t stream, bool tune_now);
}

extern "C" int dtt_gemm_batched(float* out, const float* a, const float* w, int batch, long rows, int k, int n,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  return gemm_batched_impl(out, a, w, batch, rows, k, n, workspace, workspace_bytes, static_cast<hipStream_t>(stream_), false);
}

// One-time candidate timing for a shape of dtt_gemm_batched (synchronises the stream; `out` receives the product).
extern "C" int dtt_gemm_batched_tune(float* out, const float* a, const float* w, int batch, long rows, int k, int n,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  return gemm_batched_impl(out, a, w, batch, rows, k, n, workspace, workspace_bytes, static_cast<hipStream_t>(stream_), true);
}

namespace {
int gemm_batched_impl(float* out, const float* a, const float* w, int batch, long rows, int k, int n, void* workspace,
                      size_t workspace_bytes, hipStream_t stream, bool tune_now) {
  DTT_REQUIRE(out && a && w, "gemm_batched: null pointer");
  DTT_REQUIRE(batch > 0 && rows > 0 && k > 0 && n > 0, "gemm_batched: bad shape");
  int dev = 0;
  DTT_REQUIRE(hipGetDevice(&dev) == hipSuccess, "gemm_batched: hipGetDevice failed");
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t& handle = g_handles[dev];
  if (!handle) DTT_REQUIRE(hipblasLtCreate(&handle) == HIPBLAS_STATUS_SUCCESS, "gemm_batched: hipblasLtCreate failed");
  const uint64_t ws = workspace ? workspace_bytes : 0;
  const float alpha = 1.f, beta = 0.f;
  Plan& p = g_batched_plans[std::make_tuple(dev, batch, rows, k, n)];
  if (p.ok && tune_now && !p.tuned) {
    hipblasLtMatmulDescDestroy(p.desc);
    hipblasLtMatrixLayoutDestroy(p.la); hipblasLtMatrixLayoutDestroy(p.lb); hipblasLtMatrixLayoutDestroy(p.lc);
    p = Plan();
  }
  if (!p.ok) {
    DTT_REQUIRE(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS,
                "gemm_batched: desc");
    const hipblasOperation_t op = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op, sizeof(op));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op, sizeof(op));
    DTT_REQUIRE(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_32F, n, k, n) == HIPBLAS_STATUS_SUCCESS &&
                    hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_32F, k, rows, k) == HIPBLAS_STATUS_SUCCESS &&
                    hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, n, rows, n) == HIPBLAS_STATUS_SUCCESS,
                "gemm_batched: layouts");
    const int32_t bc = batch;
    const int64_t sa = (int64_t)k * n, sb = (int64_t)rows * k, sc = (int64_t)rows * n;
    for (auto lay_stride : {std::make_pair(p.la, sa), std::make_pair(p.lb, sb), std::make_pair(p.lc, sc)}) {
      DTT_REQUIRE(hipblasLtMatrixLayoutSetAttribute(lay_stride.first, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)) ==
                          HIPBLAS_STATUS_SUCCESS &&
                      hipblasLtMatrixLayoutSetAttribute(lay_stride.first, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET,
                                                        &lay_stride.second, sizeof(lay_stride.second)) == HIPBLAS_STATUS_SUCCESS,
                  "gemm_batched: batch attributes");
    }
    hipblasLtMatmulPreference_t pref;
    DTT_REQUIRE(hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS, "gemm_batched: preference");
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    constexpr int kMaxCand = 48;
    static hipblasLtMatmulHeuristicResult_t cand[kMaxCand];  // under g_mu
    int found = 0;
    const char* tune_env = getenv("DTT_GEMM_AUTOTUNE");
    const bool tune = tune_now && !(tune_env && tune_env[0] == '0');
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(handle, p.desc, p.la, p.lb, p.lc, p.lc, pref,
                                                               tune ? kMaxCand : 1, cand, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    DTT_REQUIRE(st == HIPBLAS_STATUS_SUCCESS && found > 0, "gemm_batched: no hipBLASLt kernel for %d x %ld x %d x %d", batch,
                rows, k, n);
    int best = 0;
    if (found > 1) {   // out is the scratch: the caller's product is (re)computed right below
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float best_ms = 1e30f;
        for (int i = 0; i < found; ++i) {
          if (cand[i].state != HIPBLAS_STATUS_SUCCESS || cand[i].workspaceSize > ws) continue;
          bool good = true;
          for (int r = -1; r < 3 && good; ++r) {
            if (r == 0) (void)hipEventRecord(e0, stream);
            good = hipblasLtMatmul(handle, p.desc, &alpha, w, p.la, a, p.lb, &beta, out, p.lc, out, p.lc, &cand[i].algo,
                                   workspace, ws, stream) == HIPBLAS_STATUS_SUCCESS;
          }
          float ms = 0.f;
          if (!good || hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
              hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
            continue;
          if (ms < best_ms) { best_ms = ms; best = i; }
        }
      }
      (void)hipGetLastError();
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (getenv("DTT_GEMM_AUTOTUNE_VERBOSE"))
        fprintf(stderr, "[dtt] batched gemm %d x %ld x %d x %d: %d candidates, picked #%d\n", batch, rows, k, n, found, best);
    }
    p.heur = cand[best];
    p.ok = true;
    p.tuned = tune_now;
  }
  DTT_REQUIRE(p.heur.workspaceSize <= ws, "gemm_batched: workspace too small (%zu < %zu)", (size_t)ws,
              (size_t)p.heur.workspaceSize);
  const hipblasStatus_t st = hipblasLtMatmul(handle, p.desc, &alpha, w, p.la, a, p.lb, &beta, out, p.lc, out, p.lc,
                                             &p.heur.algo, workspace, ws, stream);
  DTT_REQUIRE(st == HIPBLAS_STATUS_SUCCESS, "gemm_batched: hipblasLtMatmul failed (%d)", (int)st);
  return 1;
}
}  // namespace

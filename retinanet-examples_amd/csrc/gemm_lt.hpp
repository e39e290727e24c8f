// gemm_lt.hpp -- 1x1 (pointwise) convolution of a channels_last activation as ONE library GEMM with the
// whole epilogue inside it:   y[p][o] = act( sum_c x[p][c] * w[o][c] + bias[o] (+ residual[p][o]) ).
//
// Host code only.  The contraction is a plain GEMM, so it goes to hipBLASLt (MFMA); what this file adds
// is the plumbing that lets the bias, the skip connection (beta * C with C != D) and the ReLU ride in
// the GEMM epilogue, which PyTorch's own matmul entry points cannot express together.  Measured on
// MI355X (tools/gemm1x1_probe.py) the separate epilogue pass after a bottleneck's last 1x1 convolution
// costs 2x the convolution itself (216 us vs 69 us at layer1) -- it is pure HBM traffic.
//
// hipBLASLt is bound at run time (dlopen + dlsym) instead of at link time: inside a PyTorch process the
// copy PyTorch ships is already mapped and must be the one that is used (two copies of one soname
// cannot coexist), and a host program without PyTorch passes the path of its own.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <mutex>
#include <tuple>
#include <vector>

#include "../../include/odtk_hip.h"

namespace odtk {
namespace lt {

struct Api {
  void *lib = nullptr;
  hipblasLtHandle_t handle = nullptr;
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
  bool ok = false;
};

inline std::mutex &mutex() {
  static std::mutex m;
  return m;
}
inline Api &api() {
  static Api a;
  return a;
}

// Bind the library (idempotent).  path == nullptr: the soname, i.e. whatever copy is already mapped.
inline int init(const char *path) {
  std::lock_guard<std::mutex> lock(mutex());
  Api &a = api();
  if (a.ok) return ODTK_OK;
  a.lib = dlopen(path && *path ? path : "libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!a.lib) a.lib = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
  if (!a.lib) return ODTK_ERR_UNSUPPORTED;
#define ODTK_LT_SYM(field, name)                                            \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, #name));       \
  if (!a.field) return ODTK_ERR_UNSUPPORTED;
  ODTK_LT_SYM(Create, hipblasLtCreate)
  ODTK_LT_SYM(LayoutCreate, hipblasLtMatrixLayoutCreate)
  ODTK_LT_SYM(LayoutDestroy, hipblasLtMatrixLayoutDestroy)
  ODTK_LT_SYM(DescCreate, hipblasLtMatmulDescCreate)
  ODTK_LT_SYM(DescDestroy, hipblasLtMatmulDescDestroy)
  ODTK_LT_SYM(DescSet, hipblasLtMatmulDescSetAttribute)
  ODTK_LT_SYM(PrefCreate, hipblasLtMatmulPreferenceCreate)
  ODTK_LT_SYM(PrefDestroy, hipblasLtMatmulPreferenceDestroy)
  ODTK_LT_SYM(PrefSet, hipblasLtMatmulPreferenceSetAttribute)
  ODTK_LT_SYM(Heuristic, hipblasLtMatmulAlgoGetHeuristic)
  ODTK_LT_SYM(Matmul, hipblasLtMatmul)
#undef ODTK_LT_SYM
  if (a.Create(&a.handle) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;
  a.ok = true;
  return ODTK_OK;
}

// One problem = one descriptor set + the algorithm picked for it.
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  bool tuned = true;     // false: chosen without timing (first seen while the stream was being captured)
};
using Key = std::tuple<int, uint64_t, uint32_t, uint32_t, int, int, int>;   // device, m, n, k, dtype, relu, residual

inline std::map<Key, Plan> &plans() {
  static std::map<Key, Plan> p;
  return p;
}

// Reproducible plans (odtk_gemm_plan_export / _import, include/odtk_hip.h): the choice made for a problem is the library's
// solution index -- what hipblaslt_ext::getIndexFromAlgo returns: the int at the head of the opaque algo struct.  An imported
// index is honoured when the heuristic offers that solution for the problem again (same library build: the candidate list is
// deterministic); then nothing is timed.  Otherwise the problem is timed as usual and counted in `pin_misses`.
using PinKey = std::tuple<uint64_t, uint32_t, uint32_t, int, int, int>;      // m, n, k, dtype, relu, residual
inline std::map<PinKey, int> &pinned() {
  static std::map<PinKey, int> p;
  return p;
}
inline int &pin_misses() {
  static int n = 0;
  return n;
}
inline int algo_index(const hipblasLtMatmulAlgo_t &algo) {
  int idx;
  std::memcpy(&idx, algo.data, sizeof idx);
  return idx;
}

// Build the descriptors and choose the algorithm: ask the heuristic for its candidates and TIME them on
// the caller's stream with the caller's buffers (the result in `y` is recomputed by the real call right
// after).  The default pick is tuned for square LLM shapes; these are tall-skinny (m up to 512 000, k and
// n 64..2048) and the measured best is often not the first.  One-off cost per shape, a few ms.
inline int make_plan(Plan *p, void *y, const void *x, const void *w, const float *bias, const void *residual,
                     uint64_t m, uint32_t n, uint32_t k, int dtype, int relu, void *workspace, size_t workspace_size,
                     hipStream_t stream) {
  Api &a = api();
  const hipDataType t = dtype == ODTK_BF16 ? HIP_R_16BF : (dtype == ODTK_F16 ? HIP_R_16F : HIP_R_32F);
  if (a.DescCreate(&p->desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;
  // row-major [rows, cols] == column-major [cols, rows]:  D^T (n x m) = W (n x k) * X^T (k x m)
  const int32_t op_t = HIPBLAS_OP_T, op_n = HIPBLAS_OP_N;
  a.DescSet(p->desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_t, sizeof op_t);
  a.DescSet(p->desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof op_n);
  const uint32_t epi = relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
  a.DescSet(p->desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi);
  const int32_t bias_t = HIP_R_32F;
  a.DescSet(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_t, sizeof bias_t);
  a.DescSet(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias);
  if (a.LayoutCreate(&p->a, t, k, n, k) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;    // W as (k x n), ld k
  if (a.LayoutCreate(&p->b, t, k, m, k) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;    // X as (k x m), ld k
  if (a.LayoutCreate(&p->c, t, n, m, n) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;    // Y / residual (n x m)

  hipblasLtMatmulPreference_t pref = nullptr;
  if (a.PrefCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return ODTK_ERR_HIP;
  const uint64_t max_ws = workspace_size;
  a.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof max_ws);
  constexpr int kAsk = 16;
  std::vector<hipblasLtMatmulHeuristicResult_t> found(kAsk);
  int n_found = 0;
  const hipblasStatus_t st = a.Heuristic(a.handle, p->desc, p->a, p->b, p->c, p->c, pref, kAsk, found.data(), &n_found);
  a.PrefDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || n_found <= 0) return ODTK_ERR_UNSUPPORTED;

  const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
  const void *c_ptr = residual ? residual : y;
  {
    const auto pin = pinned().find(PinKey{m, n, k, dtype, relu ? 1 : 0, residual ? 1 : 0});
    if (pin != pinned().end()) {                                         // an imported plan names the solution: no stopwatch
      for (int i = 0; i < n_found; ++i) {
        if (found[i].state != HIPBLAS_STATUS_SUCCESS || found[i].workspaceSize > workspace_size) continue;
        if (algo_index(found[i].algo) != pin->second) continue;
        p->algo = found[i].algo;
        p->workspace = found[i].workspaceSize;
        return ODTK_OK;
      }
      ++pin_misses();                                                    // (another library build: time the candidates as usual)
    }
  }
  // A stream that is being captured into a hipGraph cannot be timed (hipEventSynchronize would invalidate the capture):
  // a shape first seen during a capture takes the heuristic's first usable candidate and is NOT remembered as tuned -- the
  // caller (gemm_bias_act) keeps the plan only for the graph's own launches and re-plans on the next eager call.
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &capturing);
  if (capturing != hipStreamCaptureStatusNone) {
    for (int i = 0; i < n_found; ++i) {
      if (found[i].state != HIPBLAS_STATUS_SUCCESS || found[i].workspaceSize > workspace_size) continue;
      p->algo = found[i].algo;
      p->workspace = found[i].workspaceSize;
      p->tuned = false;
      return ODTK_OK;
    }
    return ODTK_ERR_UNSUPPORTED;
  }
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return ODTK_ERR_HIP;
  float best_ms = 1e30f;
  int best = -1;
  for (int i = 0; i < n_found; ++i) {
    if (found[i].state != HIPBLAS_STATUS_SUCCESS || found[i].workspaceSize > workspace_size) continue;
    auto run = [&]() {
      return a.Matmul(a.handle, p->desc, &alpha, w, p->a, x, p->b, &beta, c_ptr, p->c, y, p->c, &found[i].algo,
                      workspace, workspace_size, stream);
    };
    if (run() != HIPBLAS_STATUS_SUCCESS) continue;                       // warm-up (code object load)
    (void)hipEventRecord(e0, stream);
    bool ok = true;
    for (int r = 0; r < 3 && ok; ++r) ok = run() == HIPBLAS_STATUS_SUCCESS;
    (void)hipEventRecord(e1, stream);
    if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
    if (ms < best_ms) { best_ms = ms; best = i; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (best < 0) return ODTK_ERR_UNSUPPORTED;
  p->algo = found[best].algo;
  p->workspace = found[best].workspaceSize;
  return ODTK_OK;
}

inline int gemm_bias_act(void *y, const void *x, const void *w, const float *bias, const void *residual,
                         uint64_t m, uint32_t n, uint32_t k, int dtype, int relu, void *workspace,
                         size_t workspace_size, hipStream_t stream) {
  if (!api().ok) {
    const int rc = init(nullptr);
    if (rc != ODTK_OK) return rc;
  }
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return ODTK_ERR_HIP;
  const Key key{device, m, n, k, dtype, relu ? 1 : 0, residual ? 1 : 0};
  std::lock_guard<std::mutex> lock(mutex());     // descriptors carry the bias pointer: one call at a time
  auto it = plans().find(key);
  if (it != plans().end() && !it->second.tuned) {
    // planned blind inside a capture: time the candidates now if this call is an eager one
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &capturing);
    if (capturing == hipStreamCaptureStatusNone) {
      Api &a = api();
      a.LayoutDestroy(it->second.a); a.LayoutDestroy(it->second.b); a.LayoutDestroy(it->second.c); a.DescDestroy(it->second.desc);
      plans().erase(it);
      it = plans().end();
    }
  }
  if (it == plans().end()) {
    Plan p;
    const int rc = make_plan(&p, y, x, w, bias, residual, m, n, k, dtype, relu, workspace, workspace_size, stream);
    if (rc != ODTK_OK) {                           // nothing half-built stays behind
      Api &a = api();
      if (p.a) a.LayoutDestroy(p.a);
      if (p.b) a.LayoutDestroy(p.b);
      if (p.c) a.LayoutDestroy(p.c);
      if (p.desc) a.DescDestroy(p.desc);
      return rc;
    }
    it = plans().emplace(key, p).first;
  }
  Plan &p = it->second;
  Api &a = api();
  a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias);
  const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
  const hipblasStatus_t st = a.Matmul(a.handle, p.desc, &alpha, w, p.a, x, p.b, &beta, residual ? residual : y, p.c, y,
                                      p.c, &p.algo, workspace, workspace_size, stream);
  return st == HIPBLAS_STATUS_SUCCESS ? ODTK_OK : ODTK_ERR_HIP;
}

// "gemm m n k dtype relu residual solution-index" per tuned problem (any device).  Returns the bytes the text needs (NUL included);
// writes at most `cap` of them.
inline size_t plan_export(char *buf, size_t cap) {
  std::lock_guard<std::mutex> lock(mutex());
  std::string out;
  std::map<PinKey, int> seen;
  for (const auto &kv : plans()) {
    if (!kv.second.tuned) continue;
    const Key &q = kv.first;
    seen[PinKey{std::get<1>(q), std::get<2>(q), std::get<3>(q), std::get<4>(q), std::get<5>(q), std::get<6>(q)}] = algo_index(kv.second.algo);
  }
  for (const auto &kv : seen) {
    char line[160];
    std::snprintf(line, sizeof line, "gemm %llu %u %u %d %d %d %d\n", static_cast<unsigned long long>(std::get<0>(kv.first)),
                  std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), std::get<4>(kv.first), std::get<5>(kv.first),
                  kv.second);
    out += line;
  }
  if (buf && cap) {
    const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
    std::memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return out.size() + 1;
}

// Lines of plan_export (others are ignored): problems not planned yet in this process will take the named solution.  Returns how
// many were taken.
inline int plan_import(const char *text) {
  if (!text) return 0;
  std::lock_guard<std::mutex> lock(mutex());
  int n_in = 0;
  for (const char *p = text; *p;) {
    unsigned long long m;
    unsigned n, k;
    int dtype, relu, res, idx;
    if (std::sscanf(p, "gemm %llu %u %u %d %d %d %d", &m, &n, &k, &dtype, &relu, &res, &idx) == 7) {
      pinned()[PinKey{m, n, k, dtype, relu, res}] = idx;
      ++n_in;
    }
    const char *nl = std::strchr(p, '\n');
    if (!nl) break;
    p = nl + 1;
  }
  return n_in;
}

}  // namespace lt
}  // namespace odtk

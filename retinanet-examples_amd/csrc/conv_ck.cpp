// conv_ck.cpp -- libodtk_conv.so: k x k convolution of a channels_last activation with the bias and the ReLU in the
// convolution's OWN epilogue:   y[n][p][q][k] = act( sum_{c,r,s} x[n][p*u+r-pad][q*v+s-pad][c] * w[k][r][s][c] + bias[k] ).
//
// Host code only.  The contraction is a library convolution on the matrix cores, like the ones PyTorch / MIOpen run for the
// engine (odtk/fused.py); what this file adds is the epilogue.  MIOpen's convolution entry points cannot carry a bias + ReLU
// for bf16 NHWC (its fusion plans fall back to naive kernels: 222 ms per convolution), so every 3x3
// convolution of the engine was followed by `odtk_bias_act` -- a separate read-modify-write pass over the activation, 61
// launches and 0.56 ms of a 7.1 ms step.  composable_kernel ships the SAME implicit-GEMM kernels MIOpen
// picks for these layers (`kernel_grouped_conv_fwd_xdl_cshuffle_v3` ...) instantiated with an `AddClamp` epilogue:
//     e = clamp(acc + d, floor, ceil),   d = the bias broadcast over N, H, W  (stride 0),  ReLU = clamp(., 0, FLT_MAX)
// in /opt/rocm/lib/libdevice_conv_operations.a (`add_device_grouped_conv2d_fwd_bias_clamp_xdl_nhwgc_gkyxc_nhwgk_*`).  This
// library links those instance lists, picks the fastest instance per problem by timing them once on the caller's stream (as
// csrc/gemm_lt.hpp does for the 1x1 convolutions on hipBLASLt), and afterwards only enqueues.
//
// It is a library of its own (not part of libodtk_hip.so): the instance archive members carry code objects for a dozen GPU
// families and tens of MB, and the post-processing ABI must load without them.  The engine loads it lazily and compares it
// with the convolution + odtk_bias_act pair per layer shape; either is a native path.
//
// Reference equivalent: none (the reference runs conv -> bias / frozen BN -> ReLU as separate PyTorch kernels,
// odtk/backbones/layers.py:5-16, odtk/model.py:57-62).
#include <hip/hip_runtime.h>

#include <array>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "ck/ck.hpp"
#include "ck/library/tensor_operation_instance/gpu/grouped_convolution_forward_bias_clamp.hpp"
#include "ck/tensor_operation/gpu/device/tensor_layout.hpp"
#include "ck/tensor_operation/gpu/element/element_wise_operation.hpp"

#include "../../include/odtk_conv.h"
#include "../../include/odtk_hip.h"

namespace {

using ck::index_t;
using PassThrough = ck::tensor_operation::element_wise::PassThrough;
using AddClamp = ck::tensor_operation::element_wise::AddClamp;
namespace layout = ck::tensor_layout::convolution;

template <typename T>
using ConvOp = ck::tensor_operation::device::DeviceGroupedConvFwdMultipleABD<
    2, layout::NHWGC, layout::GKYXC, ck::Tuple<layout::NHWGK>, layout::NHWGK, T, T, ck::Tuple<T>, T, PassThrough, PassThrough,
    AddClamp, T, T>;

struct Problem {
  int n, c, h, w, k, r, s, u, v, ph, pw, ph1, pw1, dtype;   // ph / pw: padding before, ph1 / pw1: after
  bool operator<(const Problem &o) const {
    return std::tie(n, c, h, w, k, r, s, u, v, ph, pw, ph1, pw1, dtype) <
           std::tie(o.n, o.c, o.h, o.w, o.k, o.r, o.s, o.u, o.v, o.ph, o.pw, o.ph1, o.pw1, o.dtype);
  }
};

struct Plan {
  int index = -1;          // instance of the list; -1: none supports the problem
  float us = 0.0f;         // its time when it was chosen (0: chosen without timing, under a stream capture)
  bool timed = false;
  std::string name;
};

std::mutex g_mutex;
thread_local std::string g_last_plan;

template <typename T>
struct Instances {
  std::vector<std::unique_ptr<ConvOp<T>>> ops;
  std::map<Problem, Plan> plans;
  Instances() { ops = ck::tensor_operation::device::instance::DeviceOperationInstanceFactory<ConvOp<T>>::GetInstances(); }
};

template <typename T>
Instances<T> &instances() {
  static Instances<T> inst;
  return inst;
}

template <typename T>
std::unique_ptr<ck::tensor_operation::device::BaseArgument> make_argument(ConvOp<T> &op, const Problem &p, void *y, const void *x,
                                                                          const void *w, const void *bias, int relu) {
  const index_t G = 1, N = p.n, C = p.c, K = p.k, Hi = p.h, Wi = p.w, Y = p.r, X = p.s;
  const index_t Ho = (Hi + p.ph + p.ph1 - Y) / p.u + 1, Wo = (Wi + p.pw + p.pw1 - X) / p.v + 1;
  // lengths in the order CK wants them (G, N, C | K, spatial...), strides of the NHWGC / GKYXC / NHWGK memory layouts
  const std::array<index_t, 5> a_len{G, N, C, Hi, Wi}, a_str{C, Hi * Wi * G * C, 1, Wi * G * C, G * C};
  const std::array<index_t, 5> b_len{G, K, C, Y, X}, b_str{K * Y * X * C, Y * X * C, 1, X * C, C};
  const std::array<index_t, 5> e_len{G, N, K, Ho, Wo}, e_str{K, Ho * Wo * G * K, 1, Wo * G * K, G * K};
  const std::array<index_t, 5> d_str{K, 0, 1, 0, 0};                           // the bias: one value per output channel
  const std::array<index_t, 2> strides{p.u, p.v}, dilations{1, 1}, pads{p.ph, p.pw}, pads_end{p.ph1, p.pw1};
  return op.MakeArgumentPointer(x, w, std::array<const void *, 1>{bias}, y, a_len, a_str, b_len, b_str,
                                std::array<std::array<index_t, 5>, 1>{e_len}, std::array<std::array<index_t, 5>, 1>{d_str}, e_len,
                                e_str, strides, dilations, pads, pads_end, PassThrough{}, PassThrough{},
                                relu ? AddClamp{0.0f, FLT_MAX} : AddClamp{-FLT_MAX, FLT_MAX});
}

bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}

// The plan of the planned problem closest in size among those with the same channels, kernel, stride and padding, if its pixel
// count is within a factor of two of this problem's (a batch of another size is tuned on its own) and its instance takes this
// problem too.
template <typename T>
bool adopt_sibling(Instances<T> &inst, const Problem &p, void *y, const void *x, const void *w, const void *bias, int relu, Plan *out) {
  const double pixels = static_cast<double>(p.n) * p.h * p.w;
  const Plan *best = nullptr;
  double best_ratio = 2.0;
  for (const auto &kv : inst.plans) {
    const Problem &q = kv.first;
    if (kv.second.index < 0 || !kv.second.timed) continue;
    if (std::tie(q.c, q.k, q.r, q.s, q.u, q.v, q.ph, q.pw, q.ph1, q.pw1, q.dtype) !=
        std::tie(p.c, p.k, p.r, p.s, p.u, p.v, p.ph, p.pw, p.ph1, p.pw1, p.dtype))
      continue;
    const double other = static_cast<double>(q.n) * q.h * q.w;
    const double ratio = other > pixels ? other / pixels : pixels / other;
    if (ratio <= best_ratio) { best_ratio = ratio; best = &kv.second; }
  }
  if (!best) return false;
  auto &op = *inst.ops[best->index];
  auto arg = make_argument<T>(op, p, y, x, w, bias, relu);
  if (!op.IsSupportedArgument(arg.get()) || op.GetWorkSpaceSize(arg.get()) != 0) return false;
  *out = *best;
  out->us = 0.0f;                                           // (not measured on this problem)
  return true;
}

template <typename T>
int run(const Problem &p, void *y, const void *x, const void *w, const void *bias, int relu, hipStream_t stream, int force_index) {
  std::lock_guard<std::mutex> lock(g_mutex);
  Instances<T> &inst = instances<T>();
  Plan plan;
  auto it = inst.plans.find(p);
  if (it != inst.plans.end()) plan = it->second;
  const bool capturing = stream_is_capturing(stream);
  if (force_index >= 0) {
    plan = Plan{};
    if (force_index < static_cast<int>(inst.ops.size())) {
      auto arg = make_argument<T>(*inst.ops[force_index], p, y, x, w, bias, relu);
      if (inst.ops[force_index]->IsSupportedArgument(arg.get()) && inst.ops[force_index]->GetWorkSpaceSize(arg.get()) == 0)
        plan.index = force_index;
    }
    if (plan.index < 0) return ODTK_ERR_UNSUPPORTED;
    plan.name = inst.ops[plan.index]->GetTypeString();
  } else if (it == inst.plans.end() && adopt_sibling<T>(inst, p, y, x, w, bias, relu, &plan)) {
    // a problem that differs from a planned one in its extents only (a data set's batches are padded to the largest image of
    // the batch: dozens of geometries) runs on that sibling's instance -- no second tuning pass of 237 candidates per layer
    inst.plans[p] = plan;
  } else if (it == inst.plans.end() || (!plan.timed && !capturing && plan.index >= 0)) {
    // first call for this problem (or the first eager call after a capture chose blindly): time every instance that
    // supports it -- one warm-up + three timed launches each on the caller's stream, then one synchronisation per candidate
    plan = Plan{};
    struct Events {                                           // (RAII: a throwing candidate must not leak them)
      hipEvent_t e0 = nullptr, e1 = nullptr;
      ~Events() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
      }
    } ev;
    if (!capturing && (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess)) return ODTK_ERR_HIP;
    float best = 0.0f;
    for (size_t i = 0; i < inst.ops.size(); ++i) {
      try {                                                   // CK throws on arguments an instance cannot run: next candidate
        auto &op = *inst.ops[i];
        auto arg = make_argument<T>(op, p, y, x, w, bias, relu);
        if (!op.IsSupportedArgument(arg.get()) || op.GetWorkSpaceSize(arg.get()) != 0) continue;
        if (capturing) { plan.index = static_cast<int>(i); break; }          // no timing inside a capture: the first that fits
        auto invoker = op.MakeInvokerPointer();
        const StreamConfig cfg{stream, false};
        invoker->Run(arg.get(), cfg);
        if (hipEventRecord(ev.e0, stream) != hipSuccess) { (void)hipGetLastError(); continue; }
        for (int rep = 0; rep < 3; ++rep) invoker->Run(arg.get(), cfg);
        if (hipEventRecord(ev.e1, stream) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (hipEventSynchronize(ev.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev.e0, ev.e1) != hipSuccess || !(ms > 0.0f)) { (void)hipGetLastError(); continue; }   // (0 ms: a launch that did not run)
        if (plan.index < 0 || ms < best) { best = ms; plan.index = static_cast<int>(i); }
      } catch (...) {
        continue;
      }
    }
    if (plan.index >= 0) {
      plan.us = best * 1000.0f / 3.0f;
      plan.timed = !capturing;
      plan.name = inst.ops[plan.index]->GetTypeString();
    } else if (!capturing) {
      plan.timed = true;                                      // nothing takes the problem: remembered, never enumerated again
    }
    inst.plans[p] = plan;
  }
  if (plan.index < 0) return ODTK_ERR_UNSUPPORTED;
  auto &op = *inst.ops[plan.index];
  auto arg = make_argument<T>(op, p, y, x, w, bias, relu);
  op.MakeInvokerPointer()->Run(arg.get(), StreamConfig{stream, false});
  char buf[64];
  std::snprintf(buf, sizeof buf, "#%d %.1f us ", plan.index, plan.us);
  g_last_plan = std::string(buf) + plan.name;
  return hipGetLastError() == hipSuccess ? ODTK_OK : ODTK_ERR_HIP;
}

}  // namespace

namespace {
template <typename T>
void export_plans(std::string *out) {
  for (const auto &kv : instances<T>().plans) {
    const Problem &p = kv.first;
    if (kv.second.index < 0 || !kv.second.timed) continue;
    char line[256];
    std::snprintf(line, sizeof line, "conv %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d ", p.dtype, p.n, p.c, p.h, p.w, p.k, p.r, p.s, p.u, p.v,
                  p.ph, p.pw, p.ph1, p.pw1, kv.second.index);
    *out += line + kv.second.name + "\n";
  }
}
template <typename T>
bool import_plan(const Problem &p, int index, const std::string &name) {
  Instances<T> &inst = instances<T>();
  if (index < 0 || index >= static_cast<int>(inst.ops.size())) return false;
  if (inst.ops[index]->GetTypeString() != name) return false;            // another build of the instance archive: not the same kernel
  Plan plan;
  plan.index = index;
  plan.timed = true;
  plan.name = name;
  inst.plans[p] = plan;
  return true;
}
}  // namespace

extern "C" {

// declared in include/odtk_conv.h
int odtk_conv_bias_act_pads(void *y, const void *x, const void *w, const void *bias, int batch_size, int c_in, int height, int width,
                            int c_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int pad_h_end,
                            int pad_w_end, int dtype, int relu, void *stream) {
  if (!y || !x || !w || !bias || batch_size <= 0 || c_in <= 0 || c_out <= 0 || height <= 0 || width <= 0 || kernel_h <= 0 ||
      kernel_w <= 0 || stride_h <= 0 || stride_w <= 0 || pad_h < 0 || pad_w < 0 || pad_h_end < 0 || pad_w_end < 0)
    return ODTK_ERR_INVALID;
  if (height + pad_h + pad_h_end < kernel_h || width + pad_w + pad_w_end < kernel_w) return ODTK_ERR_INVALID;
  const Problem p{batch_size, c_in, height, width, c_out, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, pad_h_end, pad_w_end, dtype};
  int force = -1;
  if (const char *f = std::getenv("ODTK_CONV_INSTANCE")) force = std::atoi(f);   // A/B knob for measurements
  try {
    if (dtype == ODTK_BF16) return run<ck::bhalf_t>(p, y, x, w, bias, relu, static_cast<hipStream_t>(stream), force);
    if (dtype == ODTK_F16) return run<ck::half_t>(p, y, x, w, bias, relu, static_cast<hipStream_t>(stream), force);
  } catch (const std::exception &e) {                                            // CK throws on arguments it cannot run
    g_last_plan = std::string("exception: ") + e.what();
    return ODTK_ERR_UNSUPPORTED;
  } catch (...) {                                                                // nothing may cross the C boundary
    g_last_plan = "exception";
    return ODTK_ERR_UNSUPPORTED;
  }
  return ODTK_ERR_UNSUPPORTED;
}

int odtk_conv_bias_act(void *y, const void *x, const void *w, const void *bias, int batch_size, int c_in, int height, int width,
                       int c_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                       int relu, void *stream) {
  return odtk_conv_bias_act_pads(y, x, w, bias, batch_size, c_in, height, width, c_out, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                 pad_w, pad_h, pad_w, dtype, relu, stream);
}

const char *odtk_conv_last_plan(void) { return g_last_plan.c_str(); }

// declared in include/odtk_conv.h
size_t odtk_conv_plan_export(char *text, size_t capacity) {
  std::string out;
  try {
    std::lock_guard<std::mutex> lock(g_mutex);
    export_plans<ck::bhalf_t>(&out);
    export_plans<ck::half_t>(&out);
  } catch (...) {
    out.clear();
  }
  if (text && capacity) {
    const size_t n = out.size() < capacity - 1 ? out.size() : capacity - 1;
    std::memcpy(text, out.data(), n);
    text[n] = 0;
  }
  return out.size() + 1;
}

int odtk_conv_plan_import(const char *text) {
  if (!text) return 0;
  int taken = 0;
  try {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (const char *q = text; *q;) {
      Problem p{};
      int index = -1, used = 0;
      if (std::sscanf(q, "conv %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %n", &p.dtype, &p.n, &p.c, &p.h, &p.w, &p.k, &p.r, &p.s, &p.u,
                      &p.v, &p.ph, &p.pw, &p.ph1, &p.pw1, &index, &used) == 15 && used > 0) {
        const char *name = q + used, *end = std::strchr(name, '\n');
        const std::string nm = end ? std::string(name, end) : std::string(name);
        if (p.dtype == ODTK_BF16 ? import_plan<ck::bhalf_t>(p, index, nm) : (p.dtype == ODTK_F16 && import_plan<ck::half_t>(p, index, nm)))
          ++taken;
      }
      const char *nl = std::strchr(q, '\n');
      if (!nl) break;
      q = nl + 1;
    }
  } catch (...) {
  }
  return taken;
}

int odtk_conv_instance_count(int dtype) {
  try {
    if (dtype == ODTK_BF16) return static_cast<int>(instances<ck::bhalf_t>().ops.size());
    if (dtype == ODTK_F16) return static_cast<int>(instances<ck::half_t>().ops.size());
  } catch (...) {
  }
  return 0;
}

}  // extern "C"

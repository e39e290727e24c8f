// odtk_hip.hip -- the C ABI (include/odtk_hip.h) over the gfx950 kernels.  Single translation
// unit: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared.
//
// Host side rules: validate, lay out the workspace, fill kernel-argument structs (level tables and
// anchors travel BY VALUE in the kernarg segment: nothing is uploaded, so calls are
// hipGraph-capturable), enqueue on the caller's stream, return.  No allocation, no host sync.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/odtk_hip.h"
#include "common.hpp"
#include "epilogue.hpp"
#include "iou.hpp"
#include "nms.hpp"
#include "prefilter.hpp"
#include "select_decode.hpp"
#include "targets.hpp"
#include "loss.hpp"
#include "gemm_lt.hpp"

namespace {

thread_local char g_last_error[256] = "";
unsigned long long *g_trace = nullptr;   // odtk_debug_set_trace

int hip_fail(hipError_t e, const char *what) {
  std::snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString(e));
  return ODTK_ERR_HIP;
}
#define ODTK_HIP_TRY(expr)                                   \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);        \
  } while (0)

// ---- measurement hooks (odtk_profile_*) ------------------------------------------------------
struct EventPair { hipEvent_t start, stop; };
struct Profiler {
  std::mutex mu;
  std::atomic<unsigned> on{0};   // bit k set: kernel id k is timed (read on every launch without the lock)
  std::vector<EventPair> pending[ODTK_KERNEL_COUNT];
  std::vector<EventPair> spare;
};
Profiler g_prof;
constexpr size_t kMaxPendingEvents = 1 << 16;

struct KernelTimer {   // RAII: records start now and stop at scope exit, on `stream`
  int id; hipStream_t stream; EventPair ev; bool active = false;
  KernelTimer(int id_, hipStream_t s) : id(id_), stream(s) {
    if (!((g_prof.on.load(std::memory_order_relaxed) >> id) & 1u)) return;
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (!((g_prof.on.load(std::memory_order_relaxed) >> id) & 1u) || g_prof.pending[id].size() >= kMaxPendingEvents) return;
    if (!g_prof.spare.empty()) { ev = g_prof.spare.back(); g_prof.spare.pop_back(); }
    else if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) return;
    active = hipEventRecord(ev.start, stream) == hipSuccess;
  }
  ~KernelTimer() {
    if (!active) return;
    (void)hipEventRecord(ev.stop, stream);
    std::lock_guard<std::mutex> lock(g_prof.mu);
    g_prof.pending[id].push_back(ev);
  }
};

// One kernel launch, timed when its id is enabled.  The event pair is handed to the launch itself
// (hipExtLaunchKernelGGL): start / stop then carry the timestamps of THIS dispatch's begin and end -- the same
// clock pair rocprofv3's kernel trace reports -- instead of two separate marker packets around it, which add the
// dispatch latency on both sides (measured: 57.9 vs 52.8 us for the same prefilter launches).
template <typename K, typename... Args>
void timed_launch(int id, K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args... args) {
  if ((g_prof.on.load(std::memory_order_relaxed) >> id) & 1u) {
    EventPair ev;
    bool ok = false;
    {
      std::lock_guard<std::mutex> lock(g_prof.mu);
      if (g_prof.pending[id].size() < kMaxPendingEvents) {
        if (!g_prof.spare.empty()) { ev = g_prof.spare.back(); g_prof.spare.pop_back(); ok = true; }
        else ok = hipEventCreate(&ev.start) == hipSuccess && hipEventCreate(&ev.stop) == hipSuccess;
      }
    }
    if (ok) {
      hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), stream, ev.start, ev.stop, 0, args...);
      std::lock_guard<std::mutex> lock(g_prof.mu);
      g_prof.pending[id].push_back(ev);
      return;
    }
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, stream, args...);
}

// Kernels that want more than 64 KiB of dynamic LDS opt in with hipFuncSetAttribute -- an attribute of the function ON
// THE CURRENT DEVICE, so it is set once per (kernel, device), not once per process (a process that drives several GPUs
// would otherwise launch on the second one without it).
int allow_dynamic_lds(const void *kernel, size_t bytes, const char *what) {
  static std::mutex mu;
  static std::vector<std::pair<const void *, int>> done;
  int device = 0;
  ODTK_HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(mu);
  for (const auto &d : done)
    if (d.first == kernel && d.second == device) return ODTK_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e != hipSuccess) return hip_fail(e, what);
  done.emplace_back(kernel, device);
  return ODTK_OK;
}

constexpr size_t kAlign = 256;
inline size_t align_up(size_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

struct DecodeLayout {
  size_t sel_off, counts_off, key_off[ODTK_MAX_LEVELS], surv_off[ODTK_MAX_LEVELS], total;
  uint32_t cnt_off[ODTK_MAX_LEVELS], n[ODTK_MAX_LEVELS], spans[ODTK_MAX_LEVELS], parts[ODTK_MAX_LEVELS];
  uint32_t span_tiles, span_elems, budget;
};
static_assert(sizeof(odtk::DecodeArgs) <= 4096 && sizeof(odtk::ScanArgs) <= 4096, "kernel arguments travel by value");

// top_n <= 4096: the standard select_decode (32 KiB static sort buffer); beyond: the variant with 128 KiB of dynamic LDS
uint32_t sort_cap_for(int top_n) { return top_n <= odtk::kSortCap ? odtk::kSortCap : odtk::kSortCapBig; }

// Order of the prefilter's workgroups inside a level: consecutive workgroups take the same span of DIFFERENT images (the
// launch sweeps `batch` fronts through memory at once; measured 41.1 vs 42.0 us back to back, 48.1 vs 49.0 us in the step),
// ODTK_SCAN_ORDER=1: one image after the other (A/B measurements; the result does not depend on it)
int scan_image_major() {
  static const int v = [] { const char *e = std::getenv("ODTK_SCAN_ORDER"); return (e && e[0] == '1') ? 1 : 0; }();
  return v;
}

// select_decode's cooperative route (csrc/select_decode.hpp): how long a workgroup waits for the partners of its segment, in ticks
// of the 100 MHz wall clock, before the segment falls back to the tournament.  ODTK_SELECT_COOP_TICKS: 0 = route off (A/B),
// 1 = every barrier times out at once unless the partners are already there (exercises the fall-back), default 1000 = 10 us:
// measured where the kernel runs -- inside Model.forward, level streams on, 200 traced steps = 3200 shared segments
// (profiles/r06_select_routes_instep.txt) -- every segment went the cooperative route and workgroup 0 sat 2.1 us (p50) / 3.1 us
// (p99) / 3.4 us (max) between "slice fetched" and "barrier passed", its own histogram atomics included; the same with a 5 us
// bound.  10 us is three times the longest wait seen; a partner that is NOT resident costs its segment 10 us, not round 5's 30.
// A/B knobs of select_decode's partition (defaults = the constants of csrc/select_decode.hpp): workgroups PROVIDED per segment =
// ceil(spans / ODTK_SELECT_SPANS_PER_PART), workgroups that TAKE PART = ceil(candidates / ODTK_SELECT_KEYS_PER_PART) of them
uint32_t select_spans_per_part() {
  static const uint32_t v = [] { const char *e = std::getenv("ODTK_SELECT_SPANS_PER_PART"); const int x = e ? std::atoi(e) : 0;
                                 return x >= 1 && x <= 4096 ? static_cast<uint32_t>(x) : odtk::kSpansPerPart; }();
  return v;
}
uint32_t select_keys_per_part() {
  static const uint32_t v = [] { const char *e = std::getenv("ODTK_SELECT_KEYS_PER_PART"); const int x = e ? std::atoi(e) : 0;
                                 return x >= 64 && x <= 4096 ? static_cast<uint32_t>(x) : odtk::kKeysPerPart; }();
  return v;
}

uint32_t select_coop_ticks() {
  static const uint32_t v = [] {
    const char *e = std::getenv("ODTK_SELECT_COOP_TICKS");
    const long x = e ? std::atol(e) : 1000;
    return static_cast<uint32_t>(x < 0 ? 0 : (x > 1000000 ? 1000000 : x));
  }();
  return v;
}

// ODTK_SELECT_RANK=1: select_decode orders the selected keys by COUNTING (histogram bases + in-bin ranks) and decodes each where
// it lies (csrc/select_decode.hpp, round 6) instead of the rank-merge sort + decode loop of rounds 4-5.  Same result
// (the GPU suite passes either way); NOT the default: measured slower on the bench's bf16 heads, whose boundary score is a
// plateau of 300-500 equal 16-bit scores (37.4 vs 30.0 us in the step; profiles/r06_select_rank_by_counting.txt).
uint32_t select_rank_sort() {
  static const uint32_t v = [] { const char *e = std::getenv("ODTK_SELECT_RANK"); return (e && e[0] == '1') ? 1u : 0u; }();
  return v;
}

// axis-aligned NMS rounds: batched push (round 6, csrc/nms.hpp) or, ODTK_NMS_CHUNKS=1, the chunk loop of rounds 3-5 (A/B; same result)
bool nms_chunk_mode() {
  static const bool v = [] { const char *e = std::getenv("ODTK_NMS_CHUNKS"); return e && e[0] == '1'; }();
  return v;
}

// Tiles per prefilter workgroup for 16-bit inputs (ODTK_SCAN_SPAN = 1, 2 or 4; A/B measurements)
uint32_t scan_span_tiles() {
  static const uint32_t v = [] {
    const char *e = std::getenv("ODTK_SCAN_SPAN");
    const int x = e ? std::atoi(e) : 2;
    return static_cast<uint32_t>(x == 1 || x == 4 ? x : 2);
  }();
  return v;
}

// Workspace of a decode call: [segment state | sub-list lengths | candidate pool: kSpanCap keys per span | survivor lists].
// A span = 1 (fp32) or 2 (16-bit) tiles of 16 384 scores of ONE image; its four prefilter waves own kWaveStage keys each, so
// no list can overflow into another and nothing is reserved at run time.  A wave with more raw hits than that marks its
// list kListOverflow and select_decode re-reads the span's raw scores: capacity is a speed knob, never a result.
int decode_layout(int batch, int n_levels, const odtk_level_t *levels, int A, int C, int top_n, int dtype, DecodeLayout *out) {
  size_t off = 0;
  out->sel_off = off;
  off += align_up(sizeof(odtk::SelSeg) * static_cast<size_t>(batch) * n_levels);
  out->span_tiles = dtype == ODTK_F32 ? 1u : scan_span_tiles();
  out->span_elems = out->span_tiles * odtk::kTile;
  out->budget = sort_cap_for(top_n);                       // keys a workgroup of the tournament route passes on: one sort buffer
  size_t lists = 0;
  for (int l = 0; l < n_levels; ++l) {
    const unsigned long long n = 1ull * A * C * levels[l].height * levels[l].width;
    if (n == 0 || n > 0x7fff0000ull) return ODTK_ERR_INVALID;
    out->n[l] = static_cast<uint32_t>(n);
    out->spans[l] = static_cast<uint32_t>((n + out->span_elems - 1) / out->span_elems);
    // workgroups select_decode provides per segment: one per kSpansPerPart spans (they leave at once unless the segment
    // holds more than kKeysPerPart candidates each), and enough of them for a slice's sub-list lengths to fit in LDS
    uint32_t parts = (out->spans[l] + select_spans_per_part() - 1) / select_spans_per_part();
    if (parts > odtk::kMaxParts) parts = odtk::kMaxParts;
    const uint32_t fit = (out->spans[l] * odtk::kScanWaves + odtk::kCntSlots - 1) / odtk::kCntSlots;
    if (parts < fit) parts = fit;
    out->parts[l] = parts < 1 ? 1 : parts;
    if (lists + static_cast<size_t>(batch) * out->spans[l] * odtk::kScanWaves > 0xffffffffull) return ODTK_ERR_INVALID;
    out->cnt_off[l] = static_cast<uint32_t>(lists);
    lists += static_cast<size_t>(batch) * out->spans[l] * odtk::kScanWaves;
  }
  out->counts_off = off;
  off += align_up(sizeof(uint32_t) * lists);
  for (int l = 0; l < n_levels; ++l) {
    out->key_off[l] = off;
    off += align_up(sizeof(uint64_t) * static_cast<size_t>(batch) * out->spans[l] * odtk::kSpanCap);
  }
  for (int l = 0; l < n_levels; ++l) {
    out->surv_off[l] = off;
    off += align_up(sizeof(uint64_t) * static_cast<size_t>(batch) * out->parts[l] * out->budget);
  }
  out->total = off;
  return ODTK_OK;
}

// Conservative raw-domain prefilter for ODTK_FLAG_LOGITS: every x with score_of(x) >= thresh has
// x >= logit_lower_bound(thresh).  The margin covers the rounding of the sigmoid to bf16/f16
// (relative 2^-8) and any non-monotonicity of expf by orders of magnitude; elements that pass it
// are then tested exactly, so a looser bound only costs a few extra exp() evaluations.
float logit_lower_bound(float thresh) {
  if (!(thresh > 0.0f)) return -INFINITY;                 // sigmoid > 0 >= thresh: everything passes
  const double t = static_cast<double>(thresh) * (1.0 - 1.0 / 64.0);
  if (t >= 1.0) return 0.0f;                              // only saturated scores can pass
  return static_cast<float>(std::log(t / (1.0 - t)) - 0.01);
}

template <typename T, bool kLogits>
int launch_decode(bool rotated, bool aligned, uint32_t scan_blocks, uint32_t sel_blocks, uint32_t sort_cap, size_t scan_lds,
                  const odtk::ScanArgs &sa, const odtk::DecodeArgs &da, hipStream_t stream) {
  if (aligned)
    timed_launch(ODTK_KERNEL_PREFILTER, odtk::prefilter_scan_kernel<T, kLogits, true>, dim3(scan_blocks), dim3(odtk::kScanThreads), scan_lds, stream, sa);
  else
    timed_launch(ODTK_KERNEL_PREFILTER, odtk::prefilter_scan_kernel<T, kLogits, false>, dim3(scan_blocks), dim3(odtk::kScanThreads), scan_lds, stream, sa);
  ODTK_HIP_TRY(hipGetLastError());
  // select_decode's LDS (sort buffer, sub-histograms, sub-list lengths) is one dynamic allocation above the 64 KiB a kernel
  // gets by default: every variant opts in, once per device.  top_n > 4096 (the reference has no cap): the variant with a
  // 128 KiB sort buffer.
#define ODTK_SELECT_LAUNCH(NB_, CAP_)                                                                                          \
  do {                                                                                                                         \
    const int rc_ = allow_dynamic_lds(reinterpret_cast<const void *>(&odtk::select_decode_kernel<NB_, T, kLogits, CAP_>),       \
                                      odtk::SelLds<CAP_>::total, "hipFuncSetAttribute(select_decode_kernel)");                 \
    if (rc_ != ODTK_OK) return rc_;                                                                                            \
    timed_launch(ODTK_KERNEL_SELECT, odtk::select_decode_kernel<NB_, T, kLogits, CAP_>, dim3(sel_blocks), dim3(odtk::kSelThreads), \
                 odtk::SelLds<CAP_>::total, stream, da);                                                                       \
  } while (0)
  if (sort_cap > static_cast<uint32_t>(odtk::kSortCap)) {
    if (rotated) ODTK_SELECT_LAUNCH(6, odtk::kSortCapBig); else ODTK_SELECT_LAUNCH(4, odtk::kSortCapBig);
  } else {
    if (rotated) ODTK_SELECT_LAUNCH(6, odtk::kSortCap); else ODTK_SELECT_LAUNCH(4, odtk::kSortCap);
  }
#undef ODTK_SELECT_LAUNCH
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int decode_levels_impl(int batch, int n_levels, const odtk_level_t *levels, int A, int C, int dtype,
                       uint32_t flags, float thresh, int top_n, void *const *outputs, int n_outputs,
                       void *workspace, size_t workspace_size, hipStream_t stream, uint32_t *run_valid = nullptr) {
  if (batch <= 0 || n_levels <= 0 || n_levels > ODTK_MAX_LEVELS || !levels) return ODTK_ERR_INVALID;
  if (A <= 0 || A > ODTK_MAX_ANCHORS || C <= 0 || top_n <= 0 || top_n > ODTK_MAX_TOP_N) return ODTK_ERR_INVALID;
  for (int l = 0; l < n_levels; ++l)
    if (levels[l].height <= 0 || levels[l].width <= 0) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  size_t scan_lds = 0;                                     // per-channel threshold table of the prefilter (floats)
  for (int l = 0; l < n_levels; ++l) {
    if (levels[l].channels_last != 0 && levels[l].channels_last != 1) return ODTK_ERR_INVALID;
    if (levels[l].cls_bias) {
      if (dtype == ODTK_F32 || !(flags & ODTK_FLAG_LOGITS) || !levels[l].channels_last || (A * C) % 8 != 0)
        return ODTK_ERR_UNSUPPORTED;
      scan_lds = align_up(static_cast<size_t>(A) * C * sizeof(float));
      if (scan_lds > 48 * 1024) return ODTK_ERR_UNSUPPORTED;
      if (reinterpret_cast<uintptr_t>(levels[l].cls_thresholds) & 15u) return ODTK_ERR_INVALID;
    }
  }

  DecodeLayout lay;
  int rc = decode_layout(batch, n_levels, levels, A, C, top_n, dtype, &lay);
  if (rc != ODTK_OK) return rc;
  if (!workspace || !workspace_size) {
    if (lay.total > 0x7fffffffull) return ODTK_ERR_INVALID;   // the int return cannot carry it
    return static_cast<int>(lay.total);
  }
  if (workspace_size < lay.total) return ODTK_ERR_WORKSPACE;
  if (!outputs || n_outputs < 3 || !outputs[0] || !outputs[1] || !outputs[2]) return ODTK_ERR_INVALID;
  for (int l = 0; l < n_levels; ++l) {
    if (!levels[l].cls || !levels[l].box || !levels[l].anchors) return ODTK_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(levels[l].cls) & 15u) return ODTK_ERR_INVALID;   // 16-B vector loads
  }

  char *ws = static_cast<char *>(workspace);
  odtk::ScanArgs sa;
  std::memset(&sa, 0, sizeof sa);
  odtk::DecodeArgs da;
  std::memset(&da, 0, sizeof da);
  const uint32_t per_load = dtype == ODTK_F32 ? 4u : 8u;
  bool aligned = true;                                     // every image of every level starts on a 16-byte boundary
  uint32_t scan_blocks = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (lay.n[l] % per_load) aligned = false;
    sa.lv[l].cls = levels[l].cls;
    sa.lv[l].key_off = (lay.key_off[l] - lay.key_off[0]) / sizeof(uint64_t);
    sa.lv[l].cnt_off = lay.cnt_off[l];
    sa.lv[l].n = lay.n[l];
    sa.lv[l].blk_begin = scan_blocks;
    sa.lv[l].spans = lay.spans[l];
    sa.lv[l].seg_base = static_cast<uint32_t>(l) * batch;
    sa.lv[l].channels = static_cast<uint32_t>(A) * C;
    sa.lv[l].hw = static_cast<uint32_t>(levels[l].height) * levels[l].width;
    sa.lv[l].channels_last = static_cast<uint32_t>(levels[l].channels_last);
    sa.lv[l].by_channels = odtk::fastdiv_make(static_cast<uint32_t>(A) * C);
    sa.lv[l].bias = levels[l].cls_bias;
    sa.lv[l].table = levels[l].cls_bias ? levels[l].cls_thresholds : nullptr;
    if (static_cast<unsigned long long>(scan_blocks) + 1ull * batch * lay.spans[l] > 0x7fffffffull) return ODTK_ERR_INVALID;
    scan_blocks += static_cast<uint32_t>(batch) * lay.spans[l];

    da.lv[l].cls = levels[l].cls;
    da.lv[l].box = levels[l].box;
    da.lv[l].key_off = sa.lv[l].key_off;
    da.lv[l].surv_off = (lay.surv_off[l] - lay.surv_off[0]) / sizeof(uint64_t);
    da.lv[l].cnt_off = lay.cnt_off[l];
    da.lv[l].n = lay.n[l];
    da.lv[l].spans = lay.spans[l];
    da.lv[l].height = levels[l].height;
    da.lv[l].width = levels[l].width;
    da.lv[l].stride = static_cast<float>(levels[l].stride);
    da.lv[l].channels_last = static_cast<uint32_t>(levels[l].channels_last);
    da.lv[l].cls_bias = levels[l].cls_bias;
    da.lv[l].box_bias = levels[l].box_bias;
    std::memcpy(da.lv[l].anchors, levels[l].anchors, sizeof(float) * 4 * A);
    da.parts[l] = lay.parts[l];
    da.part_begin[l] = l == 0 ? 0u : da.part_begin[l - 1] + da.parts[l - 1] * static_cast<uint32_t>(batch);
  }
  da.part_begin[n_levels] = da.part_begin[n_levels - 1] + da.parts[n_levels - 1] * static_cast<uint32_t>(batch);
  for (int l = n_levels + 1; l <= ODTK_MAX_LEVELS; ++l) da.part_begin[l] = da.part_begin[n_levels];
  sa.counts = reinterpret_cast<uint32_t *>(ws + lay.counts_off);
  sa.cand = reinterpret_cast<uint64_t *>(ws + lay.key_off[0]);
  sa.sel = reinterpret_cast<odtk::SelSeg *>(ws + lay.sel_off);
  sa.n_levels = n_levels;
  sa.batch = batch;
  sa.span = static_cast<int>(lay.span_tiles);
  sa.image_major = scan_image_major();
  sa.thresh = thresh;
  sa.raw_lo = logit_lower_bound(thresh);

  da.sel = sa.sel;
  da.surv = reinterpret_cast<uint64_t *>(ws + lay.surv_off[0]);
  da.counts = sa.counts;
  da.cand = sa.cand;
  da.budget = lay.budget;
  da.span_elems = lay.span_elems;
  da.aligned = aligned ? 1u : 0u;
  da.coop_ticks = select_coop_ticks();
  da.keys_per_part = select_keys_per_part();
  da.rank_sort = select_rank_sort();
  da.raw_lo = sa.raw_lo;
  da.by_channels = odtk::fastdiv_make(static_cast<uint32_t>(A) * C);
  da.out_scores = static_cast<float *>(outputs[0]);
  da.out_boxes = static_cast<float *>(outputs[1]);
  da.out_classes = static_cast<float *>(outputs[2]);
  da.out_indices = n_outputs > 3 ? static_cast<int32_t *>(outputs[3]) : nullptr;
  da.run_valid = run_valid;
  da.n_levels = n_levels;
  da.batch = batch;
  da.num_anchors = A;
  da.num_classes = C;
  da.top_n = top_n;
  da.thresh = thresh;
  da.trace = g_trace;

  // Two launches, nothing to clear in front of them: the prefilter writes every sub-list length and zeroes the segment
  // state select_decode's tournament route counts in (rounds 1-3 cleared counters and histograms with a launch of their own).
  const uint32_t sel_blocks = da.part_begin[n_levels];
  const uint32_t sort_cap = sort_cap_for(top_n);
  const bool rotated = (flags & ODTK_FLAG_ROTATED) != 0, logits = (flags & ODTK_FLAG_LOGITS) != 0;
  if (dtype == ODTK_F32)
    return logits ? launch_decode<odtk::F32, true>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream)
                  : launch_decode<odtk::F32, false>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream);
  if (dtype == ODTK_BF16)
    return logits ? launch_decode<odtk::BF16, true>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream)
                  : launch_decode<odtk::BF16, false>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream);
  return logits ? launch_decode<odtk::F16, true>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream)
                : launch_decode<odtk::F16, false>(rotated, aligned, scan_blocks, sel_blocks, sort_cap, scan_lds, sa, da, stream);
}

template <int NB, bool kGlobalKeys, int kStage = 0>
int nms_launch(const odtk::NmsArgs &na, int batch, size_t lds, hipStream_t stream) {
  // opt this kernel in to the full 160 KiB of LDS (once per device)
  const int rc = allow_dynamic_lds(reinterpret_cast<const void *>(&odtk::nms_kernel<NB, kGlobalKeys, kStage>), 160 * 1024,
                                   "hipFuncSetAttribute(nms_kernel)");
  if (rc != ODTK_OK) return rc;
  timed_launch(kStage == 1 ? ODTK_KERNEL_NMS_ORDER : ODTK_KERNEL_NMS, odtk::nms_kernel<NB, kGlobalKeys, kStage>, dim3(batch),
               dim3(odtk::kNmsThreads), lds, stream, na);
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

// rotated boxes: first round in order -> pairwise suppression matrix on the whole chip -> resolve (csrc/nms.hpp)
template <bool kGlobalKeys>
int nms_rotated_staged(odtk::NmsArgs na, int batch, size_t lds, hipStream_t stream) {
  int rc = nms_launch<6, kGlobalKeys, 1>(na, batch, lds, stream);
  if (rc != ODTK_OK) return rc;
  odtk::SupArgs sa;
  sa.first_box = na.first_box; sa.first_cls = na.first_cls; sa.first_n = na.first_n; sa.sup = na.sup;
  sa.m_max = na.m_max; sa.thresh = na.thresh; sa.flags = na.flags;
  auto matrix = [&](uint32_t m_launch, uint32_t m_done, const uint32_t *done) {
    sa.m_launch = m_launch; sa.m_done = m_done; sa.done = done;
    const unsigned nblk = m_launch / 64;
    timed_launch(ODTK_KERNEL_NMS_MATRIX, odtk::rotated_sup_matrix_kernel, dim3(nblk * (nblk + 1) / 2 * (64 / odtk::kSupRows), batch),
                 dim3(odtk::kSupThreads), 0, stream, sa);
  };
  if (na.m_first >= na.m_max) {                              // the matrix is small: one step
    na.step = 0;
    matrix(na.m_max, 0, nullptr);
    ODTK_HIP_TRY(hipGetLastError());
    return nms_launch<6, kGlobalKeys, 2>(na, batch, lds, stream);
  }
  // two-step speculation: the pairs of the first m_first candidates, a resolve that stops there; only the images it did not
  // finish pay for the rest of the matrix and a second resolve (the other workgroups of those two launches leave at once)
  matrix(na.m_first, 0, nullptr);
  ODTK_HIP_TRY(hipGetLastError());
  na.step = 1;
  rc = nms_launch<6, kGlobalKeys, 2>(na, batch, lds, stream);
  if (rc != ODTK_OK) return rc;
  matrix(na.m_max, na.m_first, na.done);
  ODTK_HIP_TRY(hipGetLastError());
  na.step = 2;
  return nms_launch<6, kGlobalKeys, 2>(na, batch, lds, stream);
}

// ... and what the first of two matrix launches covers: 2.5 x detections_per_im (a detector whose boxes are well separated
// examines 1.5 .. 2.5 x as many candidates as it keeps), at least four chunks
uint32_t rotated_matrix_first(uint32_t m_max, int ndet) {
  size_t m = static_cast<size_t>(ndet) * 5 / 2;
  if (m < 256) m = 256;
  m = (m + 63) / 64 * 64;
  return m > m_max ? m_max : static_cast<uint32_t>(m);
}

// candidates of the first round the rotated suppression matrix covers: 8 x detections_per_im (the lazy pull of a typical
// image examines 1.5 .. 7 x as many candidates as it keeps), whole 64-candidate chunks, at most one round
// ... and as many as the resolve kernel can hold in LDS: it keeps the matrix (m x m / 64 words) where the polygon clip's
// columns will be once the first pair beyond the matrix is clipped (`ways` x 4 KiB: 704 candidates at 16 ways)
uint32_t rotated_matrix_rows(size_t count, int ndet, int ways) {
  size_t m = static_cast<size_t>(ndet) * 8;
  if (m > count) m = count;
  if (m > static_cast<size_t>(odtk::kNmsRound)) m = odtk::kNmsRound;
  m = (m + 63) / 64 * 64;
  const size_t room = static_cast<size_t>(ways) * odtk::kClipSlotsPerWave * sizeof(float2);
  while (m > 64 && m * (m / 64) * sizeof(uint64_t) > room) m -= 64;
  return static_cast<uint32_t>(m);
}

int nms_impl(int batch, const void *const *inputs, void *const *outputs, int n_outputs, size_t count,
             int ndet, float thresh, uint32_t flags, void *workspace, size_t workspace_size, hipStream_t stream,
             uint32_t sorted_run_len = 0, const uint32_t *run_valid = nullptr) {
  if (batch <= 0 || count == 0 || count > ODTK_MAX_NMS_COUNT_SCRATCH || ndet <= 0 || ndet > ODTK_MAX_NMS_DETECTIONS)
    return ODTK_ERR_INVALID;
  // up to ODTK_MAX_NMS_COUNT candidates per image everything is LDS-resident and the kernel needs no global scratch
  // (a token size keeps the reference's two-phase calling convention working unchanged); beyond that the key list
  // of every image lives in the workspace
  const int nb = (flags & ODTK_FLAG_ROTATED) ? 6 : 4;
  // ... or when the LDS-resident form does not fit next to a long kept list (detections_per_im in the thousands), or -- rotated
  // -- would leave fewer than 8 of the 16 waves a polygon-clip column (they are what evaluates box pairs)
  const odtk::NmsLds local(static_cast<uint32_t>(count > ODTK_MAX_NMS_COUNT ? 1 : count), ndet, nb, false);
  const bool global_keys = count > ODTK_MAX_NMS_COUNT || local.total > odtk::NmsLds::kLdsBudget || (nb == 6 && local.ways < 8);
  const size_t keys_bytes = global_keys ? align_up(sizeof(uint64_t) * static_cast<size_t>(batch) * count) : kAlign;
  // rotated: [first-round boxes | classes | counts | suppression matrix] behind the keys
  const uint32_t m_max = nb == 6 ? rotated_matrix_rows(count, ndet, odtk::NmsLds(static_cast<uint32_t>(global_keys ? 1 : count), ndet, nb, global_keys).ways) : 0u;
  const size_t off_fb = keys_bytes;
  const size_t off_fc = off_fb + (nb == 6 ? align_up(sizeof(float) * 6 * batch * m_max) : 0);
  const size_t off_fn = off_fc + (nb == 6 ? align_up(sizeof(float) * batch * m_max) : 0);
  const size_t off_fk = off_fn + (nb == 6 ? align_up(sizeof(uint32_t) * batch) : 0);
  const size_t off_fs = off_fk + (nb == 6 ? align_up(sizeof(uint64_t) * batch * odtk::kNmsRound) : 0);
  const size_t off_sup = off_fs + (nb == 6 ? align_up(sizeof(uint32_t) * 16 * batch) : 0);
  const size_t off_done = off_sup + (nb == 6 ? align_up(sizeof(uint64_t) * batch * m_max * (m_max / 64)) : 0);
  const size_t need = off_done + (nb == 6 ? align_up(sizeof(uint32_t) * batch) : 0);
  if (need > 0x7fffffffull) return ODTK_ERR_INVALID;
  if (!workspace || !workspace_size) return static_cast<int>(need);
  if (workspace_size < need) return ODTK_ERR_WORKSPACE;
  if (!inputs || !outputs || n_outputs < 3) return ODTK_ERR_INVALID;
  for (int i = 0; i < 3; ++i)
    if (!inputs[i] || !outputs[i]) return ODTK_ERR_INVALID;
  odtk::NmsArgs na;
  std::memset(&na, 0, sizeof na);
  na.scores = static_cast<const float *>(inputs[0]);
  na.boxes = static_cast<const float *>(inputs[1]);
  na.classes = static_cast<const float *>(inputs[2]);
  na.out_scores = static_cast<float *>(outputs[0]);
  na.out_boxes = static_cast<float *>(outputs[1]);
  na.out_classes = static_cast<float *>(outputs[2]);
  na.out_indices = n_outputs > 3 ? static_cast<int32_t *>(outputs[3]) : nullptr;
  na.count = static_cast<uint32_t>(count);
  na.run_len = sorted_run_len;
  na.run_valid = run_valid;
  na.ndet = ndet;
  na.thresh = thresh;
  na.flags = flags | (nms_chunk_mode() ? odtk::kNmsFlagChunks : 0u);
  na.trace = g_trace ? g_trace + 8 * 64 : nullptr;          // after the select_decode slots
  na.key_scratch = global_keys ? static_cast<uint64_t *>(workspace) : nullptr;
  const size_t lds = odtk::NmsLds(na.count, ndet, nb, global_keys).total;   // same carve-up the kernel computes
  if (lds > 160 * 1024) return ODTK_ERR_INVALID;
  if (nb == 6) {
    char *ws = static_cast<char *>(workspace);
    na.first_box = reinterpret_cast<float *>(ws + off_fb);
    na.first_cls = reinterpret_cast<float *>(ws + off_fc);
    na.first_n = reinterpret_cast<uint32_t *>(ws + off_fn);
    na.first_keys = reinterpret_cast<unsigned long long *>(ws + off_fk);
    na.first_state = reinterpret_cast<uint32_t *>(ws + off_fs);
    na.sup = reinterpret_cast<unsigned long long *>(ws + off_sup);
    na.m_max = m_max;
    na.m_first = rotated_matrix_first(m_max, ndet);
    na.done = reinterpret_cast<uint32_t *>(ws + off_done);
    return global_keys ? nms_rotated_staged<true>(na, batch, lds, stream) : nms_rotated_staged<false>(na, batch, lds, stream);
  }
  return global_keys ? nms_launch<4, true>(na, batch, lds, stream) : nms_launch<4, false>(na, batch, lds, stream);
}

template <typename T, bool kRes, bool kRelu>
int bias_act_launch(void *y, const float *bias, const void *res, uint64_t n, uint32_t channels, hipStream_t stream) {
  constexpr int per = T::kPerLoad;
  uint64_t done = 0;
  if (channels % per == 0 && n / per >= 256) {
    // fast form: grid stride (blocks * 256 lanes) must be a multiple of the row length in vectors
    const uint32_t vpr = channels / per;
    const uint64_t n_vec = n / per;                         // n is a multiple of channels, hence of per
    uint32_t g = vpr, m = 256;                              // unit = vpr / gcd(vpr, 256) blocks
    while (m) { const uint32_t r_ = g % m; g = m; m = r_; }
    const uint32_t unit = vpr / g;
    uint64_t blocks = (n_vec + 256ull * 4 - 1) / (256ull * 4);          // ~4 vectors per lane
    if (blocks > 256 * 16) blocks = 256 * 16;                            // <= 16 workgroups per CU
    blocks = (blocks + unit - 1) / unit * unit;
    timed_launch(ODTK_KERNEL_EPILOGUE, odtk::bias_act_kernel<T, kRes, kRelu>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                 stream, y, bias, res, n_vec, vpr);
    done = n_vec * per;
  }
  if (done < n) {
    uint64_t blocks = (n - done + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((odtk::bias_act_scalar_kernel<T, kRes, kRelu>), dim3(static_cast<unsigned>(blocks)), dim3(256),
                       0, stream, y, bias, res, done, n, channels);
  }
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

template <typename T>
int bias_act_typed(void *y, const float *bias, const void *res, uint64_t n, uint32_t c, bool relu, hipStream_t s) {
  if (res) return relu ? bias_act_launch<T, true, true>(y, bias, res, n, c, s) : bias_act_launch<T, true, false>(y, bias, res, n, c, s);
  return relu ? bias_act_launch<T, false, true>(y, bias, res, n, c, s) : bias_act_launch<T, false, false>(y, bias, res, n, c, s);
}

int bias_act_dispatch(void *y, const float *bias, const void *res, uint64_t n, uint32_t c, int dtype, bool relu,
                      hipStream_t s) {
  if (dtype == ODTK_F32) return bias_act_typed<odtk::F32>(y, bias, res, n, c, relu, s);
  if (dtype == ODTK_BF16) return bias_act_typed<odtk::BF16>(y, bias, res, n, c, relu, s);
  return bias_act_typed<odtk::F16>(y, bias, res, n, c, relu, s);
}

// Launch shape of the loss kernels (odtk_debug_loss_tuning; defaults = the measured best):
// workgroup size, resident workgroups per CU the logit walk is capped at, 16-byte vectors a lane loads per trip.
struct LossTuning {
  int threads, per_cu, unroll, box_blocks;
  int per_wave, window, box_rows;   // odtk_debug_loss_layout: per-wave sums (workspace form only), contiguous trips, the backward's
                                    // box-delta walk in memory order (csrc/loss.hpp LossArgs)
  int form;   // filled by loss_tuning_snapshot from g_loss_form (odtk_debug_loss_form): 1 = vectors of negatives take focal_plain
};
// [16-bit heads, fp32 heads][forward with atomics, backward, forward through a workspace] = threads, logit workgroups per
// CU and level, vectors per trip, box workgroups per level; measured with tools/loss_probe.py (profiles/r03_loss_probe.txt)
enum { kLossFwd = 0, kLossBwd = 1, kLossFwdWs = 2 };
std::mutex g_loss_tuning_mu;
// Round 6 (profiles/r06_loss_layout_probe.txt): the backward walks contiguous trips (window 1) with 256-thread workgroups and
// writes d(deltas) in memory order -- fp32 50.6 -> 42.6-43.1 us, bf16 34.3 -> 27.4 us.
// The forward through the workspace walks contiguous trips too (fp32: two vectors per trip); per-wave sums stay off (walk -1.3 us,
// reduce launch +0.8 us: nothing).
LossTuning g_loss_tuning[2][3] = {{{512, 1, 2, 64, 0, 0, 1, 0}, {256, 4, 1, 256, 0, 1, 1, 0}, {256, 4, 1, 256, 0, 1, 1, 0}},
                                  {{512, 1, 4, 64, 0, 0, 1, 0}, {256, 8, 2, 1024, 0, 1, 1, 0}, {256, 4, 2, 256, 0, 1, 1, 0}}};

// Arithmetic form of the classification walk with gamma = 2 (csrc/loss.hpp focal_plain): 0 = every element through the
// symmetric focal_term (rounds 3-4), 1 = vectors that hold no positive element and no logit beyond kPlainMax through
// focal_plain.  Same sums to ~1e-8, same gradients to ~1e-6 of the largest (both well inside the tested bars).
int g_loss_form = ODTK_LOSS_FORM_DEFAULT;

LossTuning loss_tuning_snapshot(int dtype, int which) {
  std::lock_guard<std::mutex> lock(g_loss_tuning_mu);
  LossTuning t = g_loss_tuning[dtype == ODTK_F32][which];
  t.form = g_loss_form;
  return t;
}

// fills the kernel arguments of one level; returns the number of workgroups it wants (0 on error, *rc set)
unsigned retina_loss_fill(odtk::LossArgs &la, int which, const void *cls, const void *box, const float *depth,
                          const float *box_target, int batch, int A, int C, int height, int width, int nb, int dtype,
                          int channels_last, float alpha, float gamma, float beta, double *sums, const float *g_cls,
                          const float *g_box, void *dcls, void *dbox, const LossTuning &t, int *rc) {
  const bool backward = which == kLossBwd;
  *rc = ODTK_ERR_INVALID;
  if (!cls || !box || !depth || !box_target || batch <= 0 || A <= 0 || C <= 0 || height <= 0 || width <= 0 || nb <= 0) return 0;
  if (channels_last != 0 && channels_last != 1) return 0;
  if ((reinterpret_cast<uintptr_t>(cls) | reinterpret_cast<uintptr_t>(box)) & 15u) return 0;   // 16-B vector loads
  if (backward && (!dcls || !dbox || ((reinterpret_cast<uintptr_t>(dcls) | reinterpret_cast<uintptr_t>(dbox)) & 15u))) return 0;
  const unsigned long long n = 1ull * batch * A * C * height * width;
  if (n >= (1ull << 32)) return 0;
  if (1ull * batch * A * nb * height * width >= (1ull << 32)) return 0;
  std::memset(&la, 0, sizeof la);
  la.cls = cls; la.box = box; la.depth = depth; la.box_target = box_target;
  la.acc = sums; la.g_cls = g_cls; la.g_box = g_box; la.dcls = dcls; la.dbox = dbox;
  la.batch = batch; la.num_anchors = A; la.num_classes = C; la.hw = static_cast<uint32_t>(height) * width; la.nb = nb;
  la.channels_last = channels_last;
  la.alpha = alpha; la.gamma = gamma; la.beta = beta;
  la.by_channels = odtk::fastdiv_make(static_cast<uint32_t>(A) * C);
  la.by_hw = odtk::fastdiv_make(la.hw);
  la.by_classes = odtk::fastdiv_make(C);
  la.by_anchors = odtk::fastdiv_make(A);
  const unsigned threads = t.threads, unroll = t.unroll;
  const unsigned per = dtype == ODTK_F32 ? 4u : 8u;
  // at least two trips of `unroll` vectors per lane where the level is large enough
  unsigned long long cls_blocks = (n / per + threads * unroll * 2ull - 1) / (threads * unroll * 2ull);
  if (cls_blocks < 1) cls_blocks = 1;
  // forward: every block ends in (up to) three double atomics on the SAME three words of its level, ~11 ns each when
  // they queue up (MI355X_MICROARCH.md "fanin") -- 4096 blocks cost 40 us of pure queueing per launch, and the ~2 800
  // box-delta blocks of round 2 (one cell per lane, two atomics each) cost ~30 us on their own: the forward launch
  // keeps both kinds of workgroup few (a lane walks several vectors / cells); backward has no such tail.
  // The cap is PER LEVEL: dealing one budget to the levels in proportion to their size (P3 = 3/4 of the logits) was
  // measured slower -- 768 atomics on P3's word instead of 256 (profiles/r03_loss_probe_proportional_dealing.txt).
  const unsigned long long block_cap = 256ull * t.per_cu;
  if (cls_blocks > block_cap) cls_blocks = block_cap;
  unsigned long long box_blocks = (1ull * batch * A * height * width + threads - 1) / threads;
  if (box_blocks > static_cast<unsigned>(t.box_blocks)) box_blocks = t.box_blocks;
  la.cls_blocks = static_cast<uint32_t>(cls_blocks);
  la.per_wave = (which == kLossFwdWs && t.per_wave) ? 1u : 0u;
  la.window = t.window ? 1u : 0u;
  la.box_rows = t.box_rows ? 1u : 0u;
  *rc = ODTK_OK;
  return static_cast<unsigned>(cls_blocks + box_blocks);
}

template <typename T, bool kBackward>
void retina_loss_dispatch(const odtk::LossLevelsArgs &la, unsigned total, const LossTuning &t, hipStream_t stream) {
  const dim3 grid(total), block(t.threads);
#ifdef ODTK_LOSS_ABLATIONS   // build flag of tools/loss_form_probe.py only (make ablations): never in the shipped library
  if constexpr (std::is_same_v<T, odtk::F32> && !kBackward) {
    // timing ablations of form 1 (wrong results on purpose; tools/loss_form_probe.py): fp32 forward, four vectors per trip
    if (t.form == 2) { timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, false, 4, 2>, grid, block, 0, stream, la); return; }
    if (t.form == 3) { timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, false, 4, 3>, grid, block, 0, stream, la); return; }
    if (t.form == 4) { timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, false, 4, 4>, grid, block, 0, stream, la); return; }
    if (t.form == 6) { timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, false, 4, 6>, grid, block, 0, stream, la); return; }
    if (t.form == 7) { timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, false, 4, 7>, grid, block, 0, stream, la); return; }
  }
#endif
  if (t.form) {
    switch (t.unroll) {
      case 1: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 1, 1>, grid, block, 0, stream, la); break;
      case 2: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 2, 1>, grid, block, 0, stream, la); break;
      default: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 4, 1>, grid, block, 0, stream, la); break;
    }
    return;
  }
  switch (t.unroll) {
    case 1: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 1, 0>, grid, block, 0, stream, la); break;
    case 2: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 2, 0>, grid, block, 0, stream, la); break;
    default: timed_launch(ODTK_KERNEL_LOSS, odtk::retina_loss_kernel<T, kBackward, 4, 0>, grid, block, 0, stream, la); break;
  }
}

// which: kLossFwd (atomics into `sums`, pre-zeroed), kLossBwd, kLossFwdWs (per-workgroup sums into `partial`, then the
// reduce launch writes `sums`).  With partial == nullptr and kLossFwdWs: returns the number of workgroups (size query).
int retina_loss_levels_launch(int which, int n_levels, const odtk_loss_level_t *levels, int batch, int A, int C, int nb,
                              int dtype, float alpha, float gamma, float beta, double *sums, const float *g_cls,
                              const float *g_box, double *partial, bool query, hipStream_t stream,
                              const LossTuning *tuning = nullptr) {
  if (n_levels <= 0 || n_levels > ODTK_MAX_LEVELS || !levels) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  const bool backward = which == kLossBwd;
  // ONE snapshot of the launch shape per call: the size query and the launch of the workspace form must agree even if
  // odtk_debug_loss_tuning runs on another thread in between
  const LossTuning t = tuning ? *tuning : loss_tuning_snapshot(dtype, which);
  odtk::LossLevelsArgs la;
  std::memset(&la, 0, sizeof la);
  la.n_levels = n_levels;
  unsigned total = 0;
  for (int l = 0; l < n_levels; ++l) {
    int rc;
    const unsigned blocks = retina_loss_fill(la.lv[l], which, levels[l].cls, levels[l].box, levels[l].depth, levels[l].box_target,
                                             batch, A, C, levels[l].height, levels[l].width, nb, dtype, levels[l].channels_last,
                                             alpha, gamma, beta, sums ? sums + 3 * l : nullptr, g_cls ? g_cls + l : nullptr,
                                             g_box ? g_box + l : nullptr, levels[l].dcls, levels[l].dbox, t, &rc);
    if (rc != ODTK_OK) return rc;
    la.lv[l].partial = which == kLossFwdWs ? partial : nullptr;
    la.block_begin[l] = total;
    total += blocks;
  }
  for (int l = n_levels; l <= ODTK_MAX_LEVELS; ++l) la.block_begin[l] = total;
  if (query) return static_cast<int>(total);
  if (dtype == ODTK_F32) backward ? retina_loss_dispatch<odtk::F32, true>(la, total, t, stream) : retina_loss_dispatch<odtk::F32, false>(la, total, t, stream);
  else if (dtype == ODTK_BF16) backward ? retina_loss_dispatch<odtk::BF16, true>(la, total, t, stream) : retina_loss_dispatch<odtk::BF16, false>(la, total, t, stream);
  else backward ? retina_loss_dispatch<odtk::F16, true>(la, total, t, stream) : retina_loss_dispatch<odtk::F16, false>(la, total, t, stream);
  ODTK_HIP_TRY(hipGetLastError());
  if (which == kLossFwdWs) {
    odtk::LossReduceArgs ra;
    std::memset(&ra, 0, sizeof ra);
    ra.partial = partial;
    ra.per = t.per_wave ? static_cast<uint32_t>(t.threads) / 64u : 1u;
    ra.sums = sums;
    for (int l = 0; l <= ODTK_MAX_LEVELS; ++l) ra.block_begin[l] = la.block_begin[l];
    timed_launch(ODTK_KERNEL_LOSS_REDUCE, odtk::loss_reduce_kernel, dim3(n_levels), dim3(odtk::kLossReduceThreads), 0, stream, ra);
    ODTK_HIP_TRY(hipGetLastError());
  }
  return ODTK_OK;
}

// one level = a one-entry level table through the same kernel
int retina_loss_launch(bool backward, const void *cls, const void *box, const float *depth, const float *box_target,
                       int batch, int A, int C, int height, int width, int nb, int dtype, int channels_last, float alpha,
                       float gamma, float beta, double *sums, const float *g_cls, const float *g_box, void *dcls,
                       void *dbox, hipStream_t stream) {
  odtk_loss_level_t lv;
  std::memset(&lv, 0, sizeof lv);
  lv.cls = cls; lv.box = box; lv.depth = depth; lv.box_target = box_target;
  lv.dcls = dcls; lv.dbox = dbox;
  lv.height = height; lv.width = width; lv.channels_last = channels_last;
  return retina_loss_levels_launch(backward ? kLossBwd : kLossFwd, 1, &lv, batch, A, C, nb, dtype, alpha, gamma, beta, sums, g_cls, g_box,
                                   nullptr, false, stream);
}

int decode_single(bool rotated, int batch, const void *const *inputs, void *const *outputs, size_t height,
                  size_t width, size_t scale, size_t A, size_t C, const float *anchors, size_t anchors_len,
                  float thresh, int top_n, void *workspace, size_t workspace_size, void *stream) {
  if (height == 0 || width == 0 || height > 0x7fffffff || width > 0x7fffffff || scale > 0x7fffffff)
    return ODTK_ERR_INVALID;
  if (A == 0 || A > ODTK_MAX_ANCHORS || anchors_len != 4 * A || (!anchors && workspace && workspace_size))
    return ODTK_ERR_INVALID;
  if (C == 0 || C > 0x7fffffff) return ODTK_ERR_INVALID;
  const bool query = !workspace || !workspace_size;
  if (!query && (!inputs || !inputs[0] || !inputs[1])) return ODTK_ERR_INVALID;
  odtk_level_t lv;
  std::memset(&lv, 0, sizeof lv);
  lv.cls = query ? nullptr : inputs[0];
  lv.box = query ? nullptr : inputs[1];
  lv.height = static_cast<int32_t>(height);
  lv.width = static_cast<int32_t>(width);
  lv.stride = static_cast<int32_t>(scale);
  lv.anchors = anchors;
  return decode_levels_impl(batch, 1, &lv, static_cast<int>(A), static_cast<int>(C), ODTK_F32,
                            rotated ? ODTK_FLAG_ROTATED : 0u, thresh, top_n, outputs, 3, workspace,
                            workspace_size, static_cast<hipStream_t>(stream));
}

}  // namespace

template <typename TIn>
int stem_pack_launch(const void *x, void *out, int batch, int height, int width, int channels_last, int out_dtype, hipStream_t stream) {
  const unsigned long long total = 1ull * batch * (height / 2) * (width / 2);
  unsigned long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  const odtk::FastDiv by_wo = odtk::fastdiv_make(static_cast<uint32_t>(width / 2));
  const odtk::FastDiv by_howo = odtk::fastdiv_make(static_cast<uint32_t>(height / 2) * static_cast<uint32_t>(width / 2));
  if (out_dtype == ODTK_BF16)
    timed_launch(ODTK_KERNEL_STEM_PACK, odtk::stem_pack_kernel<TIn, odtk::BF16>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, out,
                 static_cast<uint32_t>(batch), static_cast<uint32_t>(height), static_cast<uint32_t>(width), static_cast<uint32_t>(channels_last), by_wo, by_howo);
  else
    timed_launch(ODTK_KERNEL_STEM_PACK, odtk::stem_pack_kernel<TIn, odtk::F16>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, x, out,
                 static_cast<uint32_t>(batch), static_cast<uint32_t>(height), static_cast<uint32_t>(width), static_cast<uint32_t>(channels_last), by_wo, by_howo);
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

extern "C" {

const char *odtk_version(void) { return "odtk-hip 0.1 (gfx950)"; }

int odtk_abi_struct_size(int which) {
  switch (which) {
    case 0: return static_cast<int>(sizeof(odtk_level_t));
    case 1: return static_cast<int>(sizeof(odtk_snap_level_t));
    case 2: return static_cast<int>(sizeof(odtk_snap_rot_level_t));
    case 3: return static_cast<int>(sizeof(odtk_loss_level_t));
    default: return -1;
  }
}
const char *odtk_last_hip_error(void) { return g_last_error; }

int odtk_debug_set_trace(void *device_buffer) {
  g_trace = static_cast<unsigned long long *>(device_buffer);
  return ODTK_OK;
}

int odtk_debug_loss_tuning(int which, int fp32_heads, int threads, int blocks_per_cu, int unroll, int box_blocks) {
  if (which < 0 || which > 2 || threads < 64 || threads > odtk::kLossMaxThreads || threads % 64 || blocks_per_cu < 1 ||
      blocks_per_cu > 64 || (unroll != 1 && unroll != 2 && unroll != 4) || box_blocks < 1 || box_blocks > 16384)
    return ODTK_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_loss_tuning_mu);
  LossTuning &t = g_loss_tuning[fp32_heads != 0][which];
  t.threads = threads; t.per_cu = blocks_per_cu; t.unroll = unroll; t.box_blocks = box_blocks;   // (the layout switches stay)
  return ODTK_OK;
}

int odtk_debug_loss_layout(int which, int fp32_heads, int per_wave, int window, int box_rows) {
  if (which < 0 || which > 2 || (per_wave != 0 && per_wave != 1) || (window != 0 && window != 1) || (box_rows != 0 && box_rows != 1))
    return ODTK_ERR_INVALID;
  if (per_wave && which != kLossFwdWs) return ODTK_ERR_INVALID;   // per-wave sums exist in the workspace form only
  std::lock_guard<std::mutex> lock(g_loss_tuning_mu);
  LossTuning &t = g_loss_tuning[fp32_heads != 0][which];
  t.per_wave = per_wave; t.window = window; t.box_rows = box_rows;
  return ODTK_OK;
}

int odtk_debug_loss_form(int form) {
#ifdef ODTK_LOSS_ABLATIONS
  if (form < 0 || form > 7 || form == 5) return ODTK_ERR_INVALID;
#else
  if (form != 0 && form != 1) return ODTK_ERR_INVALID;   // the ablation forms (wrong sums on purpose) are not compiled in
#endif
  std::lock_guard<std::mutex> lock(g_loss_tuning_mu);
  g_loss_form = form;
  return ODTK_OK;
}

int odtk_profile_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.on.store(on < 0 ? ~0u : static_cast<unsigned>(on), std::memory_order_relaxed);
  return ODTK_OK;
}

int odtk_profile_collect(double total_ms[ODTK_KERNEL_COUNT], int launches[ODTK_KERNEL_COUNT]) {
  if (!total_ms || !launches) return ODTK_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  for (int k = 0; k < ODTK_KERNEL_COUNT; ++k) {
    total_ms[k] = 0.0;
    launches[k] = 0;
    while (!g_prof.pending[k].empty()) {                    // an event pair leaves `pending` before anything can fail
      const EventPair ev = g_prof.pending[k].back();
      g_prof.pending[k].pop_back();
      g_prof.spare.push_back(ev);
      ODTK_HIP_TRY(hipEventSynchronize(ev.stop));
      float ms = 0.0f;
      ODTK_HIP_TRY(hipEventElapsedTime(&ms, ev.start, ev.stop));
      total_ms[k] += ms;
      ++launches[k];
    }
  }
  return ODTK_OK;
}

int odtk_decode(int batch_size, const void *const *inputs, void *const *outputs, size_t height, size_t width,
                size_t scale, size_t num_anchors, size_t num_classes, const float *anchors, size_t anchors_len,
                float score_thresh, int top_n, void *workspace, size_t workspace_size, void *stream) {
  return decode_single(false, batch_size, inputs, outputs, height, width, scale, num_anchors, num_classes,
                       anchors, anchors_len, score_thresh, top_n, workspace, workspace_size, stream);
}

int odtk_decode_rotate(int batch_size, const void *const *inputs, void *const *outputs, size_t height,
                       size_t width, size_t scale, size_t num_anchors, size_t num_classes, const float *anchors,
                       size_t anchors_len, float score_thresh, int top_n, void *workspace, size_t workspace_size,
                       void *stream) {
  return decode_single(true, batch_size, inputs, outputs, height, width, scale, num_anchors, num_classes,
                       anchors, anchors_len, score_thresh, top_n, workspace, workspace_size, stream);
}

int odtk_nms(int batch_size, const void *const *inputs, void *const *outputs, size_t count,
             int detections_per_im, float nms_thresh, void *workspace, size_t workspace_size, void *stream) {
  return nms_impl(batch_size, inputs, outputs, 3, count, detections_per_im, nms_thresh, 0u, workspace,
                  workspace_size, static_cast<hipStream_t>(stream));
}

int odtk_nms_rotate(int batch_size, const void *const *inputs, void *const *outputs, size_t count,
                    int detections_per_im, float nms_thresh, void *workspace, size_t workspace_size,
                    void *stream) {
  return nms_impl(batch_size, inputs, outputs, 3, count, detections_per_im, nms_thresh, ODTK_FLAG_ROTATED,
                  workspace, workspace_size, static_cast<hipStream_t>(stream));
}

int odtk_nms_ex(int batch_size, const void *const *inputs, void *const *outputs, int n_outputs, size_t count,
                int detections_per_im, float nms_thresh, uint32_t flags, void *workspace,
                size_t workspace_size, void *stream) {
  return nms_impl(batch_size, inputs, outputs, n_outputs, count, detections_per_im, nms_thresh, flags,
                  workspace, workspace_size, static_cast<hipStream_t>(stream));
}

int odtk_nms_sorted_runs(int batch_size, const void *const *inputs, void *const *outputs, int n_outputs, size_t count,
                         int run_len, const uint32_t *run_valid, int detections_per_im, float nms_thresh, uint32_t flags,
                         void *workspace, size_t workspace_size, void *stream) {
  if (run_len <= 0 || count % static_cast<size_t>(run_len) != 0 || count / run_len > 8) return ODTK_ERR_INVALID;
  if (workspace && workspace_size && !run_valid) return ODTK_ERR_INVALID;
  return nms_impl(batch_size, inputs, outputs, n_outputs, count, detections_per_im, nms_thresh, flags, workspace, workspace_size,
                  static_cast<hipStream_t>(stream), static_cast<uint32_t>(run_len), run_valid);
}

int odtk_iou(const void *const *inputs, void *const *outputs, int num_boxes, int num_anchors, void *stream) {
  if (num_boxes < 0 || num_anchors < 0) return ODTK_ERR_INVALID;
  const long long pairs = 1ll * num_boxes * num_anchors;
  if (pairs == 0) return ODTK_OK;                            // empty side: nothing to write
  if (!inputs || !outputs || !inputs[0] || !inputs[1] || !outputs[0]) return ODTK_ERR_INVALID;
  if (pairs > 0x7fffffffll) return ODTK_ERR_INVALID;
  const int threads = 256;
  long long blocks = (pairs + threads - 1) / threads;
  if (blocks > 256 * 16) blocks = 256 * 16;                  // grid-stride beyond 16 workgroups per CU
  {
    KernelTimer t(ODTK_KERNEL_IOU, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(odtk::iou_pairs_kernel, dim3(static_cast<unsigned>(blocks)), dim3(threads), 0,
                       static_cast<hipStream_t>(stream), static_cast<const float *>(inputs[0]),
                       static_cast<const float *>(inputs[1]), static_cast<float *>(outputs[0]), num_boxes,
                       num_anchors);
  }
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_snap_to_anchors(int batch_size, const float *targets, int n_max, const float *anchors, int num_anchors,
                         int num_classes, int height, int width, int stride, float iou_background,
                         float iou_foreground, float *cls_target, float *box_target, float *depth, void *stream) {
  if (batch_size <= 0 || n_max < 0 || num_anchors <= 0 || num_anchors > ODTK_MAX_ANCHORS || num_classes <= 0 ||
      height <= 0 || width <= 0)
    return ODTK_ERR_INVALID;
  if (!anchors || !box_target || !depth || (n_max > 0 && !targets)) return ODTK_ERR_INVALID;   // cls_target may be null
  odtk::SnapArgs sa;
  std::memset(&sa, 0, sizeof sa);
  sa.targets = targets;
  sa.cls_target = cls_target;
  sa.box_target = box_target;
  sa.depth = depth;
  sa.n_max = n_max;
  sa.num_anchors = num_anchors;
  sa.num_classes = num_classes;
  sa.height = height;
  sa.width = width;
  sa.stride = static_cast<float>(stride);
  sa.iou_bg = iou_background;
  sa.iou_fg = iou_foreground;
  std::memcpy(sa.anchors, anchors, sizeof(float) * 4 * num_anchors);
  const long long cells = 1ll * num_anchors * height * width;
  const unsigned blocks = static_cast<unsigned>((cells + odtk::kSnapThreads - 1) / odtk::kSnapThreads);
  {
    KernelTimer t(ODTK_KERNEL_TARGETS, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(odtk::snap_to_anchors_kernel, dim3(blocks, batch_size), dim3(odtk::kSnapThreads), 0,
                       static_cast<hipStream_t>(stream), sa);
  }
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_snap_to_anchors_levels(int batch_size, const float *targets, int n_max, int n_levels,
                                const odtk_snap_level_t *levels, int num_anchors, int num_classes,
                                float iou_background, float iou_foreground, void *stream) {
  if (batch_size <= 0 || n_max < 0 || n_levels <= 0 || n_levels > ODTK_MAX_LEVELS || !levels || num_anchors <= 0 ||
      num_anchors > ODTK_MAX_ANCHORS || num_classes <= 0 || (n_max > 0 && !targets))
    return ODTK_ERR_INVALID;
  odtk::SnapLevelsArgs la;
  std::memset(&la, 0, sizeof la);
  la.n_levels = n_levels;
  unsigned total = 0;
  for (int l = 0; l < n_levels; ++l) {
    const odtk_snap_level_t &lv = levels[l];
    if (!lv.anchors || !lv.box_target || !lv.depth || lv.height <= 0 || lv.width <= 0) return ODTK_ERR_INVALID;
    odtk::SnapArgs &sa = la.lv[l];
    sa.targets = targets;
    sa.cls_target = lv.cls_target;
    sa.box_target = lv.box_target;
    sa.depth = lv.depth;
    sa.n_max = n_max;
    sa.num_anchors = num_anchors;
    sa.num_classes = num_classes;
    sa.height = lv.height;
    sa.width = lv.width;
    sa.stride = static_cast<float>(lv.stride);
    sa.iou_bg = iou_background;
    sa.iou_fg = iou_foreground;
    std::memcpy(sa.anchors, lv.anchors, sizeof(float) * 4 * num_anchors);
    la.block_begin[l] = total;
    const long long cells = 1ll * num_anchors * lv.height * lv.width;
    total += static_cast<unsigned>((cells + odtk::kSnapThreads - 1) / odtk::kSnapThreads);
  }
  for (int l = n_levels; l <= ODTK_MAX_LEVELS; ++l) la.block_begin[l] = total;
  timed_launch(ODTK_KERNEL_TARGETS, odtk::snap_to_anchors_levels_kernel, dim3(total, batch_size), dim3(odtk::kSnapThreads), 0,
               static_cast<hipStream_t>(stream), la);
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_snap_to_anchors_rotated_levels(int batch_size, const float *gt_axis, const float *gt_quads, const float *gt_class,
                                        int n_max, int n_levels, const odtk_snap_rot_level_t *levels, int num_anchors,
                                        int num_classes, float iou_background, float iou_foreground, void *stream) {
  if (batch_size <= 0 || n_max < 0 || n_levels <= 0 || n_levels > ODTK_MAX_LEVELS || !levels || num_anchors <= 0 ||
      num_classes <= 0 || (n_max > 0 && (!gt_axis || !gt_quads || !gt_class)))
    return ODTK_ERR_INVALID;
  odtk::SnapRotLevelsArgs la;
  std::memset(&la, 0, sizeof la);
  la.n_levels = n_levels;
  unsigned total = 0;
  for (int l = 0; l < n_levels; ++l) {
    const odtk_snap_rot_level_t &lv = levels[l];
    if (!lv.anchors_axis || !lv.anchors_quads || !lv.box_target || !lv.depth || lv.height <= 0 || lv.width <= 0) return ODTK_ERR_INVALID;
    odtk::SnapRotArgs &sa = la.lv[l];
    sa.gt_axis = gt_axis; sa.gt_quads = gt_quads; sa.gt_class = gt_class;
    sa.anchors_axis = lv.anchors_axis; sa.anchors_rot = lv.anchors_quads;
    sa.cls_target = lv.cls_target; sa.box_target = lv.box_target; sa.depth = lv.depth;
    sa.n_max = n_max; sa.num_anchors = num_anchors; sa.num_classes = num_classes;
    sa.height = lv.height; sa.width = lv.width;
    sa.stride = static_cast<float>(lv.stride);
    sa.iou_bg = iou_background; sa.iou_fg = iou_foreground;
    la.block_begin[l] = total;
    const long long cells = 1ll * num_anchors * lv.height * lv.width;
    if (cells > 0x7fffffffll) return ODTK_ERR_INVALID;
    total += static_cast<unsigned>((cells + odtk::kSnapThreads - 1) / odtk::kSnapThreads);
  }
  for (int l = n_levels; l <= ODTK_MAX_LEVELS; ++l) la.block_begin[l] = total;
  timed_launch(ODTK_KERNEL_TARGETS, odtk::snap_to_anchors_rotated_levels_kernel, dim3(total, batch_size), dim3(odtk::kSnapThreads), 0,
               static_cast<hipStream_t>(stream), la);
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_retina_loss_forward(const void *cls, const void *box, const float *depth, const float *box_target,
                             int batch_size, int num_anchors, int num_classes, int height, int width, int box_params,
                             int dtype, int channels_last, float alpha, float gamma, float beta, double *sums,
                             void *stream) {
  if (!sums) return ODTK_ERR_INVALID;
  ODTK_HIP_TRY(hipMemsetAsync(sums, 0, 3 * sizeof(double), static_cast<hipStream_t>(stream)));
  return retina_loss_launch(false, cls, box, depth, box_target, batch_size, num_anchors, num_classes, height, width,
                            box_params, dtype, channels_last, alpha, gamma, beta, sums, nullptr, nullptr, nullptr, nullptr,
                            static_cast<hipStream_t>(stream));
}

int odtk_retina_loss_backward(const void *cls, const void *box, const float *depth, const float *box_target,
                              int batch_size, int num_anchors, int num_classes, int height, int width, int box_params,
                              int dtype, int channels_last, float alpha, float gamma, float beta,
                              const float *grad_cls_sum, const float *grad_box_sum, void *dcls, void *dbox,
                              void *stream) {
  if (!dcls || !dbox) return ODTK_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(dcls) | reinterpret_cast<uintptr_t>(dbox)) & 15u) return ODTK_ERR_INVALID;
  return retina_loss_launch(true, cls, box, depth, box_target, batch_size, num_anchors, num_classes, height, width,
                            box_params, dtype, channels_last, alpha, gamma, beta, nullptr, grad_cls_sum, grad_box_sum, dcls,
                            dbox, static_cast<hipStream_t>(stream));
}

int odtk_retina_loss_levels_forward(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                    int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                    double *sums, void *stream) {
  if (!sums || n_levels <= 0 || n_levels > ODTK_MAX_LEVELS) return ODTK_ERR_INVALID;
  ODTK_HIP_TRY(hipMemsetAsync(sums, 0, 3 * sizeof(double) * n_levels, static_cast<hipStream_t>(stream)));
  return retina_loss_levels_launch(kLossFwd, n_levels, levels, batch_size, num_anchors, num_classes, box_params, dtype, alpha,
                                   gamma, beta, sums, nullptr, nullptr, nullptr, false, static_cast<hipStream_t>(stream));
}

int odtk_retina_loss_levels_forward_ws(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                       int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                       double *sums, void *workspace, size_t workspace_size, void *stream) {
  if (n_levels <= 0 || n_levels > ODTK_MAX_LEVELS) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  const LossTuning t = loss_tuning_snapshot(dtype, kLossFwdWs);
  const int blocks = retina_loss_levels_launch(kLossFwdWs, n_levels, levels, batch_size, num_anchors, num_classes, box_params,
                                               dtype, alpha, gamma, beta, nullptr, nullptr, nullptr, nullptr, true, nullptr, &t);
  if (blocks < 0) return blocks;
  const size_t need = (static_cast<size_t>(blocks) * (t.per_wave ? t.threads / 64 : 1) * 3 * sizeof(double) + 255) & ~static_cast<size_t>(255);
  if (!workspace) return static_cast<int>(need);                       // two-phase convention of the reference's plugins
  if (!sums) return ODTK_ERR_INVALID;
  if (workspace_size < need) return ODTK_ERR_WORKSPACE;
  if (reinterpret_cast<uintptr_t>(workspace) & 7u) return ODTK_ERR_INVALID;
  return retina_loss_levels_launch(kLossFwdWs, n_levels, levels, batch_size, num_anchors, num_classes, box_params, dtype, alpha,
                                   gamma, beta, sums, nullptr, nullptr, static_cast<double *>(workspace), false,
                                   static_cast<hipStream_t>(stream), &t);
}

int odtk_retina_loss_levels_backward(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                     int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                     const float *grad_cls_sums, const float *grad_box_sums, void *stream) {
  return retina_loss_levels_launch(kLossBwd, n_levels, levels, batch_size, num_anchors, num_classes, box_params, dtype, alpha,
                                   gamma, beta, nullptr, grad_cls_sums, grad_box_sums, nullptr, false, static_cast<hipStream_t>(stream));
}

int odtk_bias_act_maxpool(const void *y, const float *bias, void *out, int batch_size, int height, int width,
                          int channels, int dtype, int relu, void *stream) {
  if (!y || !bias || !out || batch_size <= 0 || height <= 0 || width <= 0 || channels <= 0) return ODTK_ERR_INVALID;
  if (dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if (channels % 8 != 0) return ODTK_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15u) return ODTK_ERR_INVALID;
  if (1ull * height * width * channels >= (1ull << 32)) return ODTK_ERR_UNSUPPORTED;   // 32-bit offsets inside one image
  const uint32_t ho = (static_cast<uint32_t>(height) + 1) / 2, wo = (static_cast<uint32_t>(width) + 1) / 2;
  const uint64_t work = static_cast<uint64_t>(batch_size) * ho * wo * (channels / 8);
  uint64_t blocks = (work + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;                              // grid-stride beyond 32 workgroups per CU
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint16_t *in = static_cast<const uint16_t *>(y);
  uint16_t *o = static_cast<uint16_t *>(out);
  odtk::PoolDivisors dv;
  dv.groups = odtk::fastdiv_make(static_cast<uint32_t>(channels / 8));
  dv.wo = odtk::fastdiv_make(wo);
  dv.ho = odtk::fastdiv_make(ho);
  const bool small = work < (1ull << 32);
#define ODTK_POOL_(T, R, S)                                                                                             \
  timed_launch(ODTK_KERNEL_POOL, odtk::bias_act_maxpool_kernel<T, R, S>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, in, \
               bias, o, static_cast<uint32_t>(batch_size), static_cast<uint32_t>(height),                             \
               static_cast<uint32_t>(width), static_cast<uint32_t>(channels), ho, wo, dv)
#define ODTK_POOL(T, R) do { if (small) ODTK_POOL_(T, R, true); else ODTK_POOL_(T, R, false); } while (0)
  if (dtype == ODTK_BF16) { if (relu) ODTK_POOL(odtk::BF16, true); else ODTK_POOL(odtk::BF16, false); }
  else { if (relu) ODTK_POOL(odtk::F16, true); else ODTK_POOL(odtk::F16, false); }
#undef ODTK_POOL_
#undef ODTK_POOL
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_prefilter_thresholds(const float *cls_bias, int channels, int dtype, float score_thresh, float *table, void *stream) {
  if (!cls_bias || !table || channels <= 0 || channels % 8 != 0) return ODTK_ERR_INVALID;
  if (dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(table) & 15u) return ODTK_ERR_INVALID;
  const float raw_thr = logit_lower_bound(score_thresh);
  const unsigned blocks = (static_cast<unsigned>(channels) + 255u) / 256u;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == ODTK_BF16)
    hipLaunchKernelGGL(odtk::prefilter_table_kernel<odtk::BF16>, dim3(blocks), dim3(256), 0, s, cls_bias, static_cast<uint32_t>(channels), raw_thr,
                       static_cast<uint32_t>(dtype), table);
  else
    hipLaunchKernelGGL(odtk::prefilter_table_kernel<odtk::F16>, dim3(blocks), dim3(256), 0, s, cls_bias, static_cast<uint32_t>(channels), raw_thr,
                       static_cast<uint32_t>(dtype), table);
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_upsample_nearest2x(const void *x, void *out, int batch_size, int height, int width, int channels, int dtype,
                            void *stream) {
  if (!x || !out || batch_size <= 0 || height <= 0 || width <= 0 || channels <= 0) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  const unsigned long long row_bytes = 1ull * channels * (dtype == ODTK_F32 ? 4 : 2);
  if (row_bytes % 16) return ODTK_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15u) return ODTK_ERR_INVALID;
  const unsigned long long groups = row_bytes / 16;
  const unsigned long long total = 4ull * batch_size * height * width * groups;
  if (total > 0xf0000000ull) return ODTK_ERR_INVALID;
  unsigned long long blocks = (total + 256ull * 4 - 1) / (256ull * 4);           // ~4 vectors per lane
  if (blocks > 256 * 16) blocks = 256 * 16;
  timed_launch(ODTK_KERNEL_UPSAMPLE, odtk::upsample_nearest2x_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
               static_cast<hipStream_t>(stream), static_cast<const odtk::vuint4 *>(x), static_cast<odtk::vuint4 *>(out),
               static_cast<uint32_t>(height), static_cast<uint32_t>(width), static_cast<uint32_t>(groups), static_cast<uint32_t>(total),
               odtk::fastdiv_make(static_cast<uint32_t>(groups)), odtk::fastdiv_make(2u * width), odtk::fastdiv_make(2u * height));
  ODTK_HIP_TRY(hipGetLastError());
  return ODTK_OK;
}

int odtk_stem_pack(const void *x, void *out, int batch_size, int height, int width, int in_dtype, int channels_last, int out_dtype,
                   void *stream) {
  if (!x || !out || batch_size <= 0 || height <= 0 || width <= 0 || (height & 1) || (width & 1)) return ODTK_ERR_INVALID;
  if (channels_last != 0 && channels_last != 1) return ODTK_ERR_INVALID;
  if (in_dtype != ODTK_F32 && in_dtype != ODTK_BF16 && in_dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if (out_dtype != ODTK_BF16 && out_dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(out) & 15u) return ODTK_ERR_INVALID;
  if (1ull * batch_size * (height / 2) * (width / 2) > 0xf0000000ull) return ODTK_ERR_INVALID;
  if (3ull * height * width >= (1ull << 32)) return ODTK_ERR_INVALID;       // 32-bit offsets inside one image
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (in_dtype == ODTK_F32) return stem_pack_launch<odtk::F32>(x, out, batch_size, height, width, channels_last, out_dtype, s);
  if (in_dtype == ODTK_BF16) return stem_pack_launch<odtk::BF16>(x, out, batch_size, height, width, channels_last, out_dtype, s);
  return stem_pack_launch<odtk::F16>(x, out, batch_size, height, width, channels_last, out_dtype, s);
}

int odtk_gemm_init(const char *hipblaslt_path) { return odtk::lt::init(hipblaslt_path); }

size_t odtk_gemm_plan_export(char *text, size_t capacity) { return odtk::lt::plan_export(text, capacity); }
int odtk_gemm_plan_import(const char *text) { return odtk::lt::plan_import(text); }
int odtk_gemm_plan_pin_misses(void) { return odtk::lt::pin_misses(); }

int odtk_gemm_bias_act(void *y, const void *x, const void *w, const float *bias, const void *residual, size_t m,
                       int n, int k, int dtype, int relu, void *workspace, size_t workspace_size, void *stream) {
  if (!y || !x || !w || !bias || n <= 0 || k <= 0 || residual == y) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
       reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(workspace)) & 15u)
    return ODTK_ERR_INVALID;
  if (m == 0) return ODTK_OK;
  KernelTimer t(ODTK_KERNEL_GEMM, static_cast<hipStream_t>(stream));
  return odtk::lt::gemm_bias_act(y, x, w, bias, residual, m, static_cast<uint32_t>(n), static_cast<uint32_t>(k), dtype,
                                 relu, workspace, workspace_size, static_cast<hipStream_t>(stream));
}

int odtk_bias_act(void *y, const float *bias, const void *residual, size_t n_pixels, int channels, int dtype,
                  int relu, void *stream) {
  if (!y || !bias || channels <= 0) return ODTK_ERR_INVALID;
  if (dtype != ODTK_F32 && dtype != ODTK_BF16 && dtype != ODTK_F16) return ODTK_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(y) & 15u) || (reinterpret_cast<uintptr_t>(residual) & 15u)) return ODTK_ERR_INVALID;
  const uint64_t n = static_cast<uint64_t>(n_pixels) * channels;
  if (n == 0) return ODTK_OK;
  return bias_act_dispatch(y, bias, residual, n, static_cast<uint32_t>(channels), dtype, relu != 0,
                           static_cast<hipStream_t>(stream));
}

int odtk_decode_levels(int batch_size, int n_levels, const odtk_level_t *levels, int num_anchors,
                       int num_classes, int dtype, uint32_t flags, float score_thresh, int top_n,
                       void *const *outputs, int n_outputs, void *workspace, size_t workspace_size,
                       void *stream) {
  return decode_levels_impl(batch_size, n_levels, levels, num_anchors, num_classes, dtype, flags,
                            score_thresh, top_n, outputs, n_outputs, workspace, workspace_size,
                            static_cast<hipStream_t>(stream));
}

int odtk_detect(int batch_size, int n_levels, const odtk_level_t *levels, int num_anchors, int num_classes,
                int dtype, uint32_t flags, float score_thresh, int top_n, float nms_thresh,
                int detections_per_im, void *const *outputs, void *workspace, size_t workspace_size,
                void *stream) {
  if (batch_size <= 0 || n_levels <= 0 || n_levels > ODTK_MAX_LEVELS || top_n <= 0) return ODTK_ERR_INVALID;
  const int nb = (flags & ODTK_FLAG_ROTATED) ? 6 : 4;
  const size_t count = static_cast<size_t>(n_levels) * top_n;
  // workspace = [decode scratch | cat scores | cat boxes | cat classes]
  const int dec = decode_levels_impl(batch_size, n_levels, levels, num_anchors, num_classes, dtype, flags,
                                     score_thresh, top_n, nullptr, 0, nullptr, 0, nullptr);
  if (dec < 0) return dec;
  const int nms_ws = nms_impl(batch_size, nullptr, nullptr, 3, count, detections_per_im, nms_thresh, flags, nullptr, 0, nullptr);
  if (nms_ws < 0) return nms_ws;
  const size_t off_s = align_up(static_cast<size_t>(dec));
  const size_t off_b = off_s + align_up(sizeof(float) * batch_size * count);
  const size_t off_c = off_b + align_up(sizeof(float) * batch_size * count * nb);
  const size_t off_n = off_c + align_up(sizeof(float) * batch_size * count);
  const size_t off_v = off_n + align_up(static_cast<size_t>(nms_ws));                // positive scores per (image, level) list
  const size_t total = off_v + align_up(sizeof(uint32_t) * batch_size * n_levels);
  if (!workspace || !workspace_size) return total > 0x7fffffffull ? ODTK_ERR_INVALID : static_cast<int>(total);
  if (workspace_size < total) return ODTK_ERR_WORKSPACE;
  if (!outputs) return ODTK_ERR_INVALID;
  char *ws = static_cast<char *>(workspace);
  void *cat[3] = {ws + off_s, ws + off_b, ws + off_c};
  uint32_t *run_valid = reinterpret_cast<uint32_t *>(ws + off_v);
  int rc = decode_levels_impl(batch_size, n_levels, levels, num_anchors, num_classes, dtype, flags, score_thresh,
                              top_n, cat, 3, workspace, static_cast<size_t>(dec), static_cast<hipStream_t>(stream), run_valid);
  if (rc != ODTK_OK) return rc;
  // the candidates are decode_levels' own output: n_levels runs of top_n, each already in NMS order, run_valid of them positive
  return nms_impl(batch_size, cat, outputs, 3, count, detections_per_im, nms_thresh, flags, ws + off_n,
                  static_cast<size_t>(nms_ws), static_cast<hipStream_t>(stream), static_cast<uint32_t>(top_n), run_valid);
}

}  // extern "C"

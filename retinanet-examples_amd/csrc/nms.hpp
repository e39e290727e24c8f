// nms.hpp -- batched greedy class-aware NMS: one 1024-thread workgroup per image, everything
// LDS-resident, work proportional to the number of boxes actually EXAMINED (not to count^2).
//
// Replaces reference steps N1-N7 (csrc/cuda/nms.cu:115-157: flag/select/sync, two radix sorts,
// nms_kernel<<<1,1024>>> with K serial __syncthreads rounds, gathers) for the whole batch in one
// launch with no host synchronisation.
//
// Semantics are the CPU path's (box.py:326-365): candidates `score > 0`, ordered score desc /
// position asc, +1 pixel IoU, a box survives iff no higher-ranked KEPT box of the same class has
// IoU > thresh with it (`!(iou <= thresh)`), stop after `ndetections` kept boxes.
// IoU arithmetic is written in box.py's operation order (-ffp-contract=off).
//
// Why this shape.  The reference (and a straightforward port) lets every kept box "push"
// suppression onto all later boxes: count x kept IoU evaluations and one barrier per kept box, even
// though only the first `ndetections` survivors are emitted.  Here the boxes are consumed lazily, in
// score order, and each candidate "pulls" against the boxes kept so far:
//   round    : the next 1024 best keys are radix-selected out of the LDS-resident key list and
//              sorted (bitonic, 1 key per thread) -- a full sort of all candidates never happens
//              unless the greedy scan really needs them all.
//   chunk    : 64 candidates (lane <-> candidate).  (1) all 16 waves test the SAME 64 candidates,
//              each against its own 1/16 slice of the kept list (a candidate's <= ndetections IoU
//              tests run on 16 threads); (2) wave 0 ANDs the 16 verdict words and resolves the
//              survivors sequentially with ballot / v_readlane -- registers only, no barrier inside.
// Typical inputs (100 detections found among the first few hundred candidates) finish in one round
// and one or two chunks; the worst case (all one class, heavy suppression) is bounded by
// count/64 chunks x (<= ndetections/16 IoU tests per thread + 2 barriers).
#pragma once

#include "common.hpp"
#include "rotated_iou.hpp"
#include "select_decode.hpp"   // radix_threshold, sort_keys_desc
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kNmsThreads = 1024;
constexpr int kNmsRound = 1024;        // keys selected + sorted per round (one per thread)
constexpr int kNmsChunk = 64;          // candidates resolved per chunk: one wave-width

struct NmsArgs {
  uint64_t *key_scratch;   // [batch, count] keys in the workspace when count > ODTK_MAX_NMS_COUNT (else unused)
  const float *scores;     // [batch, count]
  const float *boxes;      // [batch, count, NB]
  const float *classes;    // [batch, count]
  float *out_scores;       // [batch, ndet]
  float *out_boxes;        // [batch, ndet, NB]
  float *out_classes;      // [batch, ndet]
  int32_t *out_indices;    // optional [batch, ndet]
  uint32_t count;
  uint32_t run_len;        // != 0: the `count` candidates of an image are count / run_len runs, each sorted by (score desc,
                           // position asc) with its non-positive scores at the end -- what decode_levels writes.  odtk_detect
                           // sets it; the stand-alone nms entry points (arbitrary input) leave it 0.
  int ndet;
  float thresh;
  uint32_t flags;
  unsigned long long *trace;   // debug (odtk_debug_set_trace): 8 timestamps per workgroup, or null
};

// LDS carve-up shared by host (size) and device (pointers); every offset is 16-byte aligned.
struct NmsLds {
  static constexpr size_t kLdsBudget = 160 * 1024;
  size_t keys, sel, box, cls, kbox, kcls, kscore, ksrc, hist, misc, sup, clip, total;
  int ways;     // waves that take part in the pull phase
  __host__ __device__ NmsLds(uint32_t count, int ndet, int nb, bool global_keys = false) {
    auto up = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    size_t o = 0;
    keys = o;   o += global_keys ? 0 : up(static_cast<size_t>(count) * 8);
    sel = o;    o += up(kNmsRound * 8);
    box = o;    o += up(static_cast<size_t>(kNmsRound) * nb * 4);
    cls = o;    o += up(kNmsRound * 4);
    kbox = o;   o += up(static_cast<size_t>(ndet) * nb * 4);
    kcls = o;   o += up(static_cast<size_t>(ndet) * 4);
    kscore = o; o += up(static_cast<size_t>(ndet) * 4);
    ksrc = o;   o += up(static_cast<size_t>(ndet) * 4);
    hist = o;   o += up(kRadixBins * 4);
    misc = o;   o += up(112 * 4);
    sup = o;    o += up(kNmsChunk * 8);                     // suppression words of the current chunk
    // rotated IoU: one lane-private polygon region (4 KiB, rotated_iou.hpp) per wave that takes part
    // in the pull phase: as many of the 16 waves as the 160 KiB budget allows
    clip = o;
    ways = 16;
    if (nb == 6) {
      const size_t room = o < kLdsBudget ? (kLdsBudget - o) / (kClipSlotsPerWave * sizeof(float2)) : 0;
      ways = room >= 16 ? 16 : (room >= 1 ? static_cast<int>(room) : 1);
      o += static_cast<size_t>(ways) * kClipSlotsPerWave * sizeof(float2);
    }
    total = o;
  }
};

__device__ __forceinline__ float tmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float tmin(float a, float b) { return (a < b || a != a) ? a : b; }

// Does kept box m suppress the lower-ranked box j?  box.py:339 + :346-350, in its operation order.
__device__ __forceinline__ bool axis_suppresses(const float *m, const float *j, float thr) {
  // torch.max / torch.min / clamp propagate NaN
  const float x1 = tmax(j[0], m[0]), y1 = tmax(j[1], m[1]);
  const float x2 = tmin(j[2], m[2]), y2 = tmin(j[3], m[3]);
  float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
  w = w < 0.0f ? 0.0f : w;  // clamp(0)
  h = h < 0.0f ? 0.0f : h;
  const float inter = w * h;
  const float jarea = (j[2] - j[0] + 1.0f) * (j[3] - j[1] + 1.0f);
  const float marea = (m[2] - m[0] + 1.0f) * (m[3] - m[1] + 1.0f);
  // disjoint boxes (almost every same-class pair): inter == 0 makes the quotient +-0 whatever the (finite or infinite,
  // non-zero) union is, and `+-0 <= thr` holds for thr >= 0 -- same verdict as the division below, without its ~20
  // dependent instructions.  A zero or NaN union (0 / 0) and thr < 0 take the general path.
  const float both = jarea + marea;
  if (inter == 0.0f && thr >= 0.0f && both == both && both != 0.0f) return false;
  const float iou = inter / (both - inter);
  return !(iou <= thr);
}

template <int NB, bool kReject = true>
__device__ __forceinline__ bool box_suppresses(const float *m, const float *j, float thr, bool own_angle, float2 *q) {
  if constexpr (NB == 4) return axis_suppresses(m, j, thr);
  else return rotated_suppresses<kReject>(m, j, thr, own_angle, q);
}

// Does any box kept at ranks q0, q0 + step, ... (< q1) suppress candidate (jb, jc)?  Class words are
// fetched eight at a time (independent LDS reads) -- the common case is "no kept box of this class",
// and a one-read-per-trip loop would pay the LDS latency once per kept box.
template <int NB>
__device__ __forceinline__ bool pull_against_kept(const float *s_kcls, const float *s_kbox, int q0, int q1, int step,
                                                  const float *jb, float jc, bool alive, float thr, bool own_angle,
                                                  float2 *clip) {
  constexpr int kBatch = NB == 6 ? 2 : 8;        // rotated: the IoU + its reject need the registers (8 spills, 2 does not)
  for (int q = q0; q < q1 && alive; q += kBatch * step) {
    float kc[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) kc[u] = (q + u * step < q1) ? s_kcls[q + u * step] : __builtin_nanf("");   // NaN equals nothing
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (alive && kc[u] == jc) {                               // box.py:351: a different class keeps
        float mb[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) mb[k] = s_kbox[(q + u * step) * NB + k];
        if (box_suppresses<NB>(mb, jb, thr, own_angle, clip)) alive = false;
      }
    }
  }
  return alive;
}

struct LdsKeySource {   // keys of this image that rank below `upper` (exclusive)
  const uint64_t *keys;
  uint32_t count;
  uint64_t upper;
  template <typename F>
  __device__ __forceinline__ void for_each(F &&f) const {
    for (uint32_t i = threadIdx.x; i < count; i += kNmsThreads) {
      const uint64_t k = keys[i];
      if (k < upper) f(k);
    }
  }
};

// kGlobalKeys: more candidates than the LDS holds (count > ODTK_MAX_NMS_COUNT): the key list of an image lives in the
// caller's workspace instead; rounds then walk it out of L2 -- slower, same result.
template <int NB, bool kGlobalKeys = false>
__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const NmsLds lay(a.count, a.ndet, NB, kGlobalKeys);
  uint64_t *s_keys = kGlobalKeys ? a.key_scratch + static_cast<size_t>(blockIdx.x) * a.count
                                 : reinterpret_cast<uint64_t *>(smem + lay.keys);
  uint64_t *s_sel = reinterpret_cast<uint64_t *>(smem + lay.sel);
  float *s_box = reinterpret_cast<float *>(smem + lay.box);
  float *s_cls = reinterpret_cast<float *>(smem + lay.cls);
  float *s_kbox = reinterpret_cast<float *>(smem + lay.kbox);
  float *s_kcls = reinterpret_cast<float *>(smem + lay.kcls);
  float *s_kscore = reinterpret_cast<float *>(smem + lay.kscore);
  int32_t *s_ksrc = reinterpret_cast<int32_t *>(smem + lay.ksrc);
  uint32_t *s_hist = reinterpret_cast<uint32_t *>(smem + lay.hist);
  uint32_t *s_misc = reinterpret_cast<uint32_t *>(smem + lay.misc);
  // s_misc: [0..31] radix_select scratch, [32] key count, [33] gather cursor, [34] kept count,
  //         [40..71] verdict words of the pull phase (16 x 64 bits); [40..103] min / max keys per wave before the first round
  //         (generic mode); [72..111] per-run valid counts, cursors, member counts, probe keys (sorted-run mode)
  uint64_t *s_alive = reinterpret_cast<uint64_t *>(s_misc + 40);
  uint64_t *s_sup = reinterpret_cast<uint64_t *>(smem + lay.sup);
  float2 *s_clip = reinterpret_cast<float2 *>(smem + lay.clip);     // rotated only
  const int ways = lay.ways;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int img = blockIdx.x;
  if (a.trace && tid == 0) a.trace[(gridDim.x + blockIdx.x) * 8 + 0] = wall_clock64();
  const long long shader_clock0 = a.trace ? clock64() : 0;   // debug: shader cycles vs the 100 MHz wall clock = effective clock
  const uint32_t count = a.count;
  const int ndet = a.ndet;
  const float thr = a.thresh;
  const bool own_angle = (a.flags & ODTK_FLAG_ROTATED_NMS_FIXED_ANGLE) != 0;
  const float *in_s = a.scores + static_cast<size_t>(img) * count;
  const float *in_b = a.boxes + static_cast<size_t>(img) * count * NB;
  const float *in_c = a.classes + static_cast<size_t>(img) * count;

  // ---- compact positive-score candidates into 64-bit (score, ~position) keys ----
  if (tid == 0) { s_misc[32] = 0; s_misc[34] = 0; }
  __syncthreads();
  // Sorted-run mode (odtk_detect): the input is what decode_levels wrote -- n_runs lists of run_len candidates, each already
  // in NMS order.  The k best candidates overall are then prefixes of the runs: a round is found by looking at `step` slots
  // per run (below) and neither the key list, nor the min / max pass, nor the radix selection are needed (measured before:
  // compaction 1.8 + selection 10.8 us of the kernel's 36).
  const uint32_t n_runs = a.run_len ? count / a.run_len : 0;
  static_assert(ODTK_MAX_LEVELS <= 8, "the sorted-run state (s_misc[72..111]) holds 8 runs; run_len is set by odtk_detect only");
  const bool runs = a.run_len >= 64 && n_runs * a.run_len == count && n_runs >= 1 && n_runs <= 8;   // block-uniform
  uint32_t *s_valid = s_misc + 72, *s_cursor = s_misc + 80, *s_members = s_misc + 88;   // per run (s_misc[72..95])
  uint64_t *s_probe = reinterpret_cast<uint64_t *>(s_misc + 96);                       // per run (s_misc[96..111])
  // adds `n` to counter[run] for the lanes with pred, one LDS atomic per (wave, run): a wave's 64 consecutive slots touch
  // at most two runs (run_len, step >= 64)
  auto count_per_run = [&](uint32_t *counter, bool pred, uint32_t run) {
    const uint64_t m = __ballot(pred);
    if (!m) return;
    const uint32_t r0 = __shfl(run, __ffsll(static_cast<unsigned long long>(m)) - 1, kWave);
    const uint64_t m0 = __ballot(pred && run == r0), m1 = m & ~m0;
    if (lane == 0) {
      atomicAdd(&counter[r0], static_cast<uint32_t>(__popcll(m0)));
      if (m1) atomicAdd(&counter[r0 + 1], static_cast<uint32_t>(__popcll(m1)));
    }
  };
  if (runs) {
    if (tid < 8) { s_valid[tid] = 0; s_cursor[tid] = 0; }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < count; i0 += kNmsThreads) {
      const uint32_t i = i0 + tid;
      const float sc = i < count ? in_s[i] : 0.0f;
      count_per_run(s_valid, sc > 0.0f, i < count ? i / a.run_len : 0);   // box.py:328  score > 0 (NaN fails)
    }
    __syncthreads();
    if (tid == 0) { uint32_t k = 0; for (uint32_t l = 0; l < n_runs; ++l) k += s_valid[l]; s_misc[32] = k; }
  } else {
    auto compact = [&](float sc, uint32_t i) {                // wave-uniform call sites
      const bool pos = sc > 0.0f;                             // box.py:328  score > 0 (NaN fails)
      const uint64_t m = __ballot(pos);
      if (m) {                                                // wave-uniform
        uint32_t wbase = 0;
        if (lane == 0) wbase = atomicAdd(&s_misc[32], static_cast<uint32_t>(__popcll(m)));
        wbase = __shfl(wbase, 0, kWave);
        if (pos) s_keys[wbase + __popcll(m & ((1ull << lane) - 1ull))] = make_key(sc, i);
      }
    };
    if constexpr (kGlobalKeys) {
      for (uint32_t i0 = 0; i0 < count; i0 += kNmsThreads) {
        const uint32_t i = i0 + tid;
        compact(i < count ? in_s[i] : 0.0f, i);
      }
    } else {
      constexpr int kScoreLoads = (ODTK_MAX_NMS_COUNT + kNmsThreads - 1) / kNmsThreads;   // 8
      float my_scores[kScoreLoads];
  #pragma unroll
      for (int u = 0; u < kScoreLoads; ++u) {                   // all score loads in flight at once
        const uint32_t i = u * kNmsThreads + tid;
        my_scores[u] = i < count ? in_s[i] : 0.0f;
      }
  #pragma unroll
      for (int u = 0; u < kScoreLoads; ++u) {
        if (u * kNmsThreads >= count) break;                    // block-uniform
        compact(my_scores[u], u * kNmsThreads + tid);
      }
    }
  }
  __syncthreads();
  const uint32_t K = __builtin_amdgcn_readfirstlane(s_misc[32]);   // (LDS values are VGPRs to the compiler: pin loop-control scalars to SGPRs)
  auto stamp = [&](int k) { if (a.trace && tid == 0) a.trace[(gridDim.x + blockIdx.x) * 8 + k] = wall_clock64(); };
  stamp(1);

  // smallest / largest key of the image: the round selection below cuts THAT range (fp32 scores of one image share their
  // exponent bits; an MSD digit needed two passes, 9.9 us, to isolate the first 256..1024 keys)
  uint64_t k_lo = ~0ull, k_hi = 0;
  if (!runs) {                       // (block-uniform; the sorted-run mode needs neither and keeps its state in s_misc[72..])
  for (uint32_t i = tid; i < K; i += kNmsThreads) {
    const uint64_t k = s_keys[i];
    k_lo = k < k_lo ? k : k_lo;
    k_hi = k > k_hi ? k : k_hi;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t o1 = shfl_xor_u64(k_lo, d), o2 = shfl_xor_u64(k_hi, d);
    k_lo = o1 < k_lo ? o1 : k_lo;
    k_hi = o2 > k_hi ? o2 : k_hi;
  }
  if (lane == 0) { s_alive[wave] = k_lo; s_alive[16 + wave] = k_hi; }   // (s_misc[40..103]: the verdict words are not in use yet)
  __syncthreads();
  for (int w = 0; w < kNmsThreads / kWave; ++w) {
    k_lo = s_alive[w] < k_lo ? s_alive[w] : k_lo;
    k_hi = s_alive[16 + w] > k_hi ? s_alive[16 + w] : k_hi;
  }
  __syncthreads();
  }

  uint32_t consumed = 0;             // candidates handed to earlier rounds
  uint64_t upper = ~0ull;            // keys >= upper were consumed
  int kept = 0;                      // block-uniform copy of s_misc[34]

  while (consumed < K && kept < ndet) {
    // ---- round: select + sort the next (up to) 1024 best keys ----
    const uint32_t left = K - consumed;
    uint32_t n_round = left;
    uint64_t lower = 0;
    if (runs) {
      // slot t = (run l, offset j): the next `step` candidates of every run.  T = the largest of the runs' LAST examined keys;
      // the members of the round are the slots with key >= T: no run can hold an unexamined key >= T (its last examined
      // key is <= T and the run is sorted), so they are exactly the best unconsumed candidates, 1 .. n_runs * step of them.
      const uint32_t step = kNmsRound / n_runs;
      const uint32_t l = static_cast<uint32_t>(tid) / step, j = static_cast<uint32_t>(tid) - l * step;
      uint64_t key = 0;
      if (tid < 8) { s_members[tid] = 0; s_probe[tid] = 0; }
      if (tid == 0) s_misc[33] = 0;
      __syncthreads();
      uint32_t avail = 0;
      if (l < n_runs) {
        avail = s_valid[l] - s_cursor[l];
        if (j < avail) {
          const uint32_t p = l * a.run_len + s_cursor[l] + j;
          key = make_key(in_s[p], p);
          if (j == (avail < step ? avail : step) - 1) s_probe[l] = key;
        }
      }
      __syncthreads();
      uint64_t T = 0;
      for (uint32_t q = 0; q < n_runs; ++q) T = s_probe[q] > T ? s_probe[q] : T;
      const bool take = key != 0 && key >= T;
      const uint32_t slot = wave_append_slot(&s_misc[33], take);
      if (take) s_sel[slot] = key;
      count_per_run(s_members, take, l < n_runs ? l : 0);
      __syncthreads();
      if (static_cast<uint32_t>(tid) < n_runs) s_cursor[tid] += s_members[tid];
      n_round = s_misc[33];
      __syncthreads();
    } else {
      const LdsKeySource src{s_keys, K, upper};
      // any top-prefix of 256..1024 keys will do for a round: stop the radix descent early
      if (left > kNmsRound)
        lower = range_threshold(src, 256, kNmsRound, k_lo, upper == ~0ull ? k_hi : upper - 1, s_hist, s_misc, &n_round);
      if (tid == 0) s_misc[33] = 0;
      __syncthreads();
      for (uint32_t i0 = 0; i0 < K; i0 += kNmsThreads) {     // (block-uniform trip count: the append ballots)
        const uint32_t i = i0 + tid;
        const uint64_t key = i < K ? s_keys[i] : 0;
        const bool take = key != 0 && key < upper && key >= lower;
        const uint32_t slot = wave_append_slot(&s_misc[33], take);
        if (take) s_sel[slot] = key;
      }
      __syncthreads();
    }
    if (consumed == 0) stamp(2);
    n_round = __builtin_amdgcn_readfirstlane(n_round);
    sort_keys_desc(s_sel, n_round);                        // pads to 1024 with zeros (sort last)
    if (consumed == 0) stamp(3);
    upper = lower;
    consumed += n_round;

    // thread t <-> rank `t` of this round: stage its box + class in LDS
    if (static_cast<uint32_t>(tid) < n_round) {
      const uint32_t p = key_index(s_sel[tid]);
#pragma unroll
      for (int k = 0; k < NB; ++k) s_box[tid * NB + k] = in_b[static_cast<size_t>(p) * NB + k];
      s_cls[tid] = in_c[p];
    }
    __syncthreads();

    // ---- chunks of 64 candidates: 16-way parallel pull, then one wave resolves the 64 in order ----
    for (uint32_t c0 = 0; c0 < n_round && kept < ndet; c0 += kNmsChunk) {
      const int kept_before = kept;
      // debug: per-chunk timeline of image 0's first round (3 stamps per chunk: start, after the pull phase, after the resolve)
      auto cstamp = [&](int k) {
        if (a.trace && tid == 0 && blockIdx.x == 0 && consumed == n_round && c0 / kNmsChunk < 20)
          a.trace[2048 - 8 * 64 + (c0 / kNmsChunk) * 4 + k] = k == 3 ? static_cast<unsigned long long>(kept) : wall_clock64();
      };
      cstamp(0);
      // (1) EVERY wave looks at the same 64 candidates (lane <-> candidate).
      //   pull: against its own slice of the kept list (ranks wave, wave + ways, ...): a candidate's <= ndet tests are
      //         spread over `ways` threads.  Wave w publishes its verdict word; the AND is the set still alive.
      //   suppression rows: row i of the chunk = the lanes j > i of the same class that candidate i WOULD suppress if it is
      //         kept.  Wave w computes rows w, w + ways, ...: all of the chunk's pairwise IoUs happen here, in parallel on
      //         every wave, and not on the serial chain of (2) -- measured before: 0.28 us per kept box in (2), 28 of the
      //         kernel's 50 us (one wave, ~10 dependent branches per box).
      const uint32_t r = c0 + lane;                           // rank inside the round (< 1024)
      float jb[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) jb[k] = s_box[r * NB + k];
      const float jc = s_cls[r];
      float2 *clip = s_clip + static_cast<size_t>(wave) * kClipSlotsPerWave + lane;   // rotated: wave-private
      const uint32_t n_chunk = n_round - c0 < kNmsChunk ? n_round - c0 : kNmsChunk;
      if (wave < ways) {
        bool alive = r < n_round;
        if (alive) alive = pull_against_kept<NB>(s_kcls, s_kbox, wave, kept_before, ways, jb, jc, true, thr, own_angle, clip);
        const uint64_t word = __ballot(alive);
        if (lane == 0) s_alive[wave] = word;
        for (uint32_t i = wave; i < n_chunk; i += ways) {
          const float ic = s_cls[c0 + i];                      // same address in every lane: LDS broadcast
          const bool rival = static_cast<uint32_t>(lane) > i && r < n_round && jc == ic;
          uint64_t row = 0;
          if (__ballot(rival)) {                               // wave-uniform: most rows have no same-class follower
            float ib[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) ib[k] = s_box[(c0 + i) * NB + k];
            row = __ballot(rival && box_suppresses<NB>(ib, jb, thr, own_angle, clip));
          }
          if (lane == 0) s_sup[i] = row;
        }
      }
      __syncthreads();
      cstamp(1);

      // (2) wave 0 resolves the chunk in rank order with scalar bit operations only: lane i holds row i, the first alive
      //     candidate is kept and clears its row's bits from the alive mask (v_readlane -> s_andn2).  No IoU, no LDS, one
      //     branch per kept box.
      if (wave == 0) {
        uint64_t word = ~0ull;
        for (int w = 0; w < ways; ++w) word &= s_alive[w];
        const uint64_t my_row = static_cast<uint32_t>(lane) < n_chunk ? s_sup[lane] : 0;
        const uint32_t row_lo = static_cast<uint32_t>(my_row), row_hi = static_cast<uint32_t>(my_row >> 32);
        // (LDS values are VGPRs, "divergent" to the compiler: pin the loop state to SGPRs so the loop is scalar control flow)
        uint64_t mask = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(word >> 32))) << 32) |
                        static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(word)));
        uint64_t kept_mask = 0;
        int k_cnt = kept_before;
        while (mask && k_cnt < ndet) {
          const int l0 = __ffsll(static_cast<unsigned long long>(mask)) - 1;
          const uint64_t row = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(row_hi), l0))) << 32) |
                               static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(row_lo), l0));
          kept_mask |= 1ull << l0;
          mask &= ~(row | (1ull << l0));
          ++k_cnt;
        }
        if ((kept_mask >> lane) & 1ull) {                       // all lanes kept in this chunk write their entry, in parallel
          const int my_rank = kept_before + __popcll(kept_mask & ((1ull << lane) - 1ull));
          const uint64_t key = s_sel[r];
#pragma unroll
          for (int k = 0; k < NB; ++k) s_kbox[my_rank * NB + k] = jb[k];
          s_kcls[my_rank] = jc;
          s_kscore[my_rank] = key_score(key);
          s_ksrc[my_rank] = static_cast<int32_t>(key_index(key));
        }
        if (lane == 0) s_misc[34] = static_cast<uint32_t>(k_cnt);
      }
      __syncthreads();
      kept = __builtin_amdgcn_readfirstlane(static_cast<int>(s_misc[34]));
      cstamp(2);
      cstamp(3);
    }
  }

  stamp(4);
  if (a.trace && tid == 0) {
    a.trace[(gridDim.x + blockIdx.x) * 8 + 5] = consumed;
    a.trace[(gridDim.x + blockIdx.x) * 8 + 6] = K;
    a.trace[(gridDim.x + blockIdx.x) * 8 + 7] = static_cast<unsigned long long>(clock64() - shader_clock0);
  }
  // ---- outputs: kept boxes, then the zero-padded tail (box.py:322-324) ----
  for (int t = tid; t < ndet; t += kNmsThreads) {
    const size_t o = static_cast<size_t>(img) * ndet + t;
    const bool v = t < kept;
    a.out_scores[o] = v ? s_kscore[t] : 0.0f;
    a.out_classes[o] = v ? s_kcls[t] : 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) a.out_boxes[o * NB + k] = v ? s_kbox[t * NB + k] : 0.0f;
    if (a.out_indices) a.out_indices[o] = v ? s_ksrc[t] : -1;
  }
}

}  // namespace odtk

// nms.hpp -- batched greedy class-aware NMS: one 1024-thread workgroup per image, every
// candidate LDS-resident, at most `detections_per_im` iterations.
//
// Replaces reference steps N1-N7 (csrc/cuda/nms.cu:115-157: flag/select/sync, two radix sorts,
// nms_kernel<<<1,1024>>> with K serial __syncthreads rounds, gathers) for the whole batch in one
// launch with no host synchronisation.  The reference kernel runs K (<= 5000) barrier rounds per
// image although only the first `detections_per_im` survivors are emitted (nms.cu:49-79, :150);
// the CPU path (box.py:342-361) stops after `ndetections` kept boxes -- so does this kernel: one
// barrier round per KEPT box.
//
// Semantics are the CPU path's (box.py:326-365): candidates `score > 0`, ordered score desc /
// position asc, +1 pixel IoU, box j (after i) survives iff IoU(i,j) <= thresh or class differs.
// IoU arithmetic is written in box.py's operation order (-ffp-contract=off).
//
// Structure
//   phase 1  positive-score candidates are compacted (wave ballot + one LDS atomic per wave) into
//            64-bit (score, ~position) keys and sorted by a bitonic network over pow2(K) keys.
//   phase 2  thread t owns sorted positions t, t+1024, ...: box / class / area live in REGISTERS;
//            a copy of box + class goes to LDS (overlaying the keys) for broadcast reads.
//   phase 3  per kept box: every wave finds the first alive position from a 128-word LDS bitmap
//            (redundantly -- no extra barrier), tests its own boxes against it, publishes its
//            bitmap words into the other buffer, ONE barrier.  No global memory traffic inside
//            the loop (a global store before a barrier costs a full memory round trip per
//            iteration); the owners write the outputs after the loop.
//
// LDS plan (dynamic, <= 160 KiB): sort keys 8 B x pow2(count) are overlaid, after the sort, by
// the sorted boxes (16|24 B each) + classes (4 B each); a 2 x 128-word alive bitmap follows.
#pragma once

#include "common.hpp"
#include "rotated_iou.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kNmsThreads = 1024;
constexpr int kNmsSlots = 8;                        // sorted positions per thread (8192 max)
constexpr int kNmsWords = kNmsThreads * kNmsSlots / 64;   // 128 alive words

struct NmsArgs {
  const float *scores;     // [batch, count]
  const float *boxes;      // [batch, count, NB]
  const float *classes;    // [batch, count]
  float *out_scores;       // [batch, ndet]
  float *out_boxes;        // [batch, ndet, NB]
  float *out_classes;      // [batch, ndet]
  int32_t *out_indices;    // optional [batch, ndet]
  uint32_t count;
  uint32_t n_pow2;         // pow2 >= count
  int ndet;
  float thresh;
  uint32_t flags;
};

template <int NB>
struct BoxT { float v[NB]; };

__device__ __forceinline__ float tmax(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float tmin(float a, float b) { return (a < b || a != a) ? a : b; }

// IoU of the reference's CPU path, box.py:339 + :346-350, in its operation order.
__device__ __forceinline__ bool axis_suppresses(const float *m, float marea, const float *j, float jarea, float thr) {
  // torch.max / torch.min / clamp propagate NaN
  const float x1 = tmax(j[0], m[0]), y1 = tmax(j[1], m[1]);
  const float x2 = tmin(j[2], m[2]), y2 = tmin(j[3], m[3]);
  float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
  w = w < 0.0f ? 0.0f : w;  // clamp(0)
  h = h < 0.0f ? 0.0f : h;
  const float inter = w * h;
  const float iou = inter / (jarea + marea - inter);
  return !(iou <= thr);
}

template <int NB>
__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *s_keys = reinterpret_cast<uint64_t *>(smem);                       // phase 1
  float *s_box = reinterpret_cast<float *>(smem);                              // phase 2 (overlay)
  float *s_cls = s_box + static_cast<size_t>(a.count) * NB;
  const size_t overlay = static_cast<size_t>(a.count) * (NB + 1) * 4;
  const size_t keys_b = static_cast<size_t>(a.n_pow2) * 8;
  const size_t bitmap_off = ((overlay > keys_b ? overlay : keys_b) + 15) & ~static_cast<size_t>(15);
  uint64_t *s_alive = reinterpret_cast<uint64_t *>(smem + bitmap_off);         // [2][kNmsWords]
  uint32_t *s_cnt = reinterpret_cast<uint32_t *>(s_alive + 2 * kNmsWords);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int img = blockIdx.x;
  const uint32_t count = a.count;
  const float *in_s = a.scores + static_cast<size_t>(img) * count;
  const float *in_b = a.boxes + static_cast<size_t>(img) * count * NB;
  const float *in_c = a.classes + static_cast<size_t>(img) * count;

  // ---- phase 1: compact positive-score candidates into keys, sort descending ----
  if (tid == 0) *s_cnt = 0;
  __syncthreads();
  for (uint32_t base = 0; base < count; base += kNmsThreads) {
    const uint32_t i = base + tid;
    const float s = i < count ? in_s[i] : 0.0f;
    const bool pos = s > 0.0f;                              // box.py:328  score > 0 (NaN fails)
    const uint64_t m = __ballot(pos);
    if (m) {                                                // wave-uniform
      uint32_t wbase = 0;
      if (lane == 0) wbase = atomicAdd(s_cnt, static_cast<uint32_t>(__popcll(m)));
      wbase = __shfl(wbase, 0, kWave);
      if (pos) s_keys[wbase + __popcll(m & ((1ull << lane) - 1ull))] = make_key(s, i);
    }
  }
  __syncthreads();
  const uint32_t K = *s_cnt;
  uint32_t n_sort = 1;
  while (n_sort < K) n_sort <<= 1;
  for (uint32_t i = K + tid; i < n_sort; i += kNmsThreads) s_keys[i] = 0;   // pad: sorts last
  __syncthreads();
  for (uint32_t k = 2; k <= n_sort; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (n_sort >> 1); t += kNmsThreads) {
        const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t hi = lo | j;
        const uint64_t x = s_keys[lo], y = s_keys[hi];
        const bool desc = (lo & k) == 0;
        if (desc ? (x < y) : (x > y)) { s_keys[lo] = y; s_keys[hi] = x; }
      }
      __syncthreads();
    }
  }

  // ---- phase 2: each thread owns sorted positions p = s*1024 + tid; registers keep its boxes ----
  float r_score[kNmsSlots], r_cls[kNmsSlots], r_area[kNmsSlots];
  BoxT<NB> r_box[kNmsSlots];
  int32_t r_src[kNmsSlots], r_rank[kNmsSlots];
  uint32_t alive = 0;   // bit s: position s*1024+tid is a candidate that is neither kept nor suppressed
#pragma unroll
  for (int s = 0; s < kNmsSlots; ++s) {
    const uint32_t p = s * kNmsThreads + tid;
    r_score[s] = 0.0f; r_cls[s] = 0.0f; r_area[s] = 0.0f; r_src[s] = -1; r_rank[s] = -1;
#pragma unroll
    for (int k = 0; k < NB; ++k) r_box[s].v[k] = 0.0f;
    if (p < K) {
      const uint64_t key = s_keys[p];
      const uint32_t src = key_index(key);
      r_src[s] = static_cast<int32_t>(src);
      r_score[s] = in_s[src];
      r_cls[s] = in_c[src];
#pragma unroll
      for (int k = 0; k < NB; ++k) r_box[s].v[k] = in_b[static_cast<size_t>(src) * NB + k];
      // box.py:339  areas = (x2 - x1 + 1) * (y2 - y1 + 1)
      r_area[s] = (r_box[s].v[2] - r_box[s].v[0] + 1.0f) * (r_box[s].v[3] - r_box[s].v[1] + 1.0f);
      alive |= 1u << s;
    }
  }
  __syncthreads();   // every key has been consumed: the overlay may be written
#pragma unroll
  for (int s = 0; s < kNmsSlots; ++s) {
    const uint32_t p = s * kNmsThreads + tid;
    if (p < K) {
#pragma unroll
      for (int k = 0; k < NB; ++k) s_box[static_cast<size_t>(p) * NB + k] = r_box[s].v[k];
      s_cls[p] = r_cls[s];
    }
  }
  // alive bitmap: word (s*16 + wave) bit lane  <->  position s*1024 + wave*64 + lane
  const uint32_t n_slots = (K + kNmsThreads - 1) / kNmsThreads;   // slots that can be alive
#pragma unroll
  for (int s = 0; s < kNmsSlots; ++s) {
    const uint64_t word = __ballot((alive >> s) & 1u);
    if (lane == 0) { s_alive[s * 16 + wave] = word; s_alive[kNmsWords + s * 16 + wave] = 0; }
  }
  __syncthreads();

  // ---- phase 3: one barrier round per kept box, LDS traffic only ----
  int kept = 0;
  int buf = 0;
  const int ndet = a.ndet;
  while (kept < ndet) {
    // every wave finds the first alive position redundantly (no extra barrier)
    const uint64_t w0 = s_alive[buf * kNmsWords + lane];
    const uint64_t w1 = s_alive[buf * kNmsWords + 64 + lane];
    const uint64_t nz0 = __ballot(w0 != 0), nz1 = __ballot(w1 != 0);
    if ((nz0 | nz1) == 0) break;
    uint32_t widx; uint64_t wval;
    if (nz0) { widx = __ffsll(static_cast<unsigned long long>(nz0)) - 1; wval = __shfl(w0, widx, kWave); }
    else { widx = __ffsll(static_cast<unsigned long long>(nz1)) - 1; wval = __shfl(w1, widx, kWave); widx += 64; }
    const uint32_t bit = __ffsll(static_cast<unsigned long long>(wval)) - 1;
    // word widx = s*16 + w  ->  position s*1024 + w*64 + bit
    const uint32_t ms = widx >> 4, mw = widx & 15;
    const uint32_t m = ms * kNmsThreads + mw * 64 + bit;
    const uint32_t m_tid = mw * 64 + bit;

    // the owner retires box m and remembers its output rank (written out after the loop)
    if (static_cast<uint32_t>(tid) == m_tid) {
#pragma unroll
      for (int s = 0; s < kNmsSlots; ++s)
        if (static_cast<uint32_t>(s) == ms) r_rank[s] = kept;
      alive &= ~(1u << ms);
    }
    ++kept;
    if (kept == ndet) break;

    // everyone tests its still-alive later boxes against box m (LDS broadcast reads)
    float mb[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) mb[k] = s_box[static_cast<size_t>(m) * NB + k];
    const float mcls = s_cls[m];
    const float marea = (mb[2] - mb[0] + 1.0f) * (mb[3] - mb[1] + 1.0f);
#pragma unroll
    for (int s = 0; s < kNmsSlots; ++s) {
      if ((alive >> s) & 1u) {
        const uint32_t p = s * kNmsThreads + tid;
        if (p > m && r_cls[s] == mcls) {          // box.py:351  classes != classes[i] keeps
          bool sup;
          if (NB == 4) sup = axis_suppresses(mb, marea, r_box[s].v, r_area[s], a.thresh);
          else sup = rotated_suppresses(mb, r_box[s].v, a.thresh, (a.flags & ODTK_FLAG_ROTATED_NMS_FIXED_ANGLE) != 0);
          if (sup) alive &= ~(1u << s);
        }
      }
    }
    // publish the new bitmap into the other buffer
    buf ^= 1;
#pragma unroll
    for (int s = 0; s < kNmsSlots; ++s) {
      if (static_cast<uint32_t>(s) < n_slots) {
        const uint64_t word = __ballot((alive >> s) & 1u);
        if (lane == 0) s_alive[buf * kNmsWords + s * 16 + wave] = word;
      }
    }
    __syncthreads();
  }

  // ---- outputs: kept boxes by their owners, then the zero-padded tail (box.py:322-324) ----
#pragma unroll
  for (int s = 0; s < kNmsSlots; ++s) {
    if (r_rank[s] >= 0) {
      const size_t o = static_cast<size_t>(img) * ndet + r_rank[s];
      a.out_scores[o] = r_score[s];
      a.out_classes[o] = r_cls[s];
#pragma unroll
      for (int k = 0; k < NB; ++k) a.out_boxes[o * NB + k] = r_box[s].v[k];
      if (a.out_indices) a.out_indices[o] = r_src[s];
    }
  }
  for (int t = kept + tid; t < ndet; t += kNmsThreads) {
    const size_t o = static_cast<size_t>(img) * ndet + t;
    a.out_scores[o] = 0.0f;
    a.out_classes[o] = 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) a.out_boxes[o * NB + k] = 0.0f;
    if (a.out_indices) a.out_indices[o] = -1;
  }
}

}  // namespace odtk

// select_decode.hpp -- kernel 2 of the decode path: one 1024-thread workgroup per
// (level, image) segment picks the top_n candidates (score desc, flat index asc), sorts them
// in LDS and decodes their boxes.
//
// Replaces reference steps D4-D6 (csrc/cuda/decode.cu:108-167: gather + cub radix sort of all
// survivors, the box-decode device lambda, the tail fills) for all images and levels at once.
// Box arithmetic follows odtk/box.py:97-111 + :302 operation by operation (CPU path is
// normative: two-sided clamp).
//
// Selection is exact for ANY input:
//   K > sort size           : MULTI-WORKGROUP narrowing first (select_pass_kernel, two launches in front of this
//                             kernel): a histogram pass over 2048 equal bins of the key range, then a filter pass (with a
//                             second histogram digit inside the same launch when saturated scores need one), each walked
//                             by up to kSelParts workgroups per segment, leave the <= ~top_n keys at or above the
//                             boundary bin in a small survivor list; this kernel then only sorts.  (One workgroup
//                             walking 1e5..1e7 keys three times was the whole cost of this kernel.)
//   K <= kSortCap           : all candidates are sorted (bitonic network in LDS).
//   kSortCap < K <= cap     : MSD radix descent (11-bit digits, LDS histograms) on the 64-bit
//                             keys of the candidate lists narrows down the bin of the top_n-th
//                             key until "everything >= that bin" fits the LDS sort buffer
//                             (usually 2 passes); those keys are gathered and sorted.
//   K > cap (list overflow) : the same radix-select runs over the segment's RAW scores
//                             (keys rebuilt on the fly), so correctness never depends on cap.
// Keys are unique (they embed the index), so "keys >= T" is exactly top_n elements even when
// every score is equal.
#pragma once

#include "common.hpp"
#include "prefilter.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kSelThreads = 1024;
constexpr int kSortCap = 4096;                  // keys sortable in LDS by the standard kernel (32 KiB)
constexpr int kSortCapBig = ODTK_MAX_TOP_N;     // ... by the top_n > 4096 variant (128 KiB of dynamic LDS)
constexpr int kRadixBits = 11;
constexpr int kRadixBins = 1 << kRadixBits;

struct DecodeLevel {
  const void *cls;
  const void *box;
  uint64_t cand_off;
  uint32_t n;            // A*C*H*W
  uint32_t cap;          // per sub-list
  int32_t height, width;
  float stride;
  uint32_t channels_last;
  const float *cls_bias; // [A*C] added to the cls head values (logits) before the sigmoid, or null
  const float *box_bias; // [A*NB] added to the gathered deltas, or null
  float anchors[ODTK_MAX_ANCHORS * 4];
};

constexpr int kSelParts = 64;             // workgroups per segment of the multi-workgroup passes (largest levels)
constexpr int kPassThreads = 256;          // threads of a pass workgroup: 4 waves, so that the ~10^3 workgroups of a launch are
                                           // all resident at once (1024-thread workgroups needed two rounds: +6 us per pass)
constexpr uint32_t kSelSlice = 2048;       // candidate keys per workgroup of a pass (list source): 8 per lane, one round of loads
constexpr uint32_t kSurvCap = 16384;       // survivor keys per segment the filter pass may emit (4 x that for top_n > 4096)
constexpr uint32_t kRankCap = 1024;        // keys of the boundary bin select_decode can rank by brute force (one per thread)

// Per-segment scratch of the multi-workgroup selection; zeroed by the host memset before every call.
struct SelSeg {
  uint32_t hist[2][1 << 11];               // histograms of pass 0 / pass 1, bins reversed (largest keys first)
  unsigned long long kmax;                 // pass 0: largest key, and ...
  unsigned long long kmin_inv;             // ... largest ~key (= ~smallest key)
  // filter pass (part 0): keys >= T survive, `expected` of them; those > bin_hi (exactly `taken` keys) are wanted
  // outright, of the `in_bin` keys inside [T, bin_hi] the `need` largest are wanted
  unsigned long long T, bin_hi;
  uint32_t expected, need, n_above, n_bin;   // expected = n_above + n_bin
  uint32_t filtered, surv_count;
  uint32_t arrived, pad_;                  // second digit: workgroups of the segment that have added their histogram (ticket)
};

struct DecodeArgs {
  DecodeLevel lv[ODTK_MAX_LEVELS];
  uint32_t part_begin[ODTK_MAX_LEVELS + 1];   // first workgroup of each level in a select_pass_kernel launch
  uint32_t parts[ODTK_MAX_LEVELS];            // workgroups per segment of that level (0: level too small to need any)
  SelSeg *sel;                                // [n_levels * batch]
  uint64_t *surv;                             // [n_levels * batch][surv_cap]
  uint32_t sort_cap, surv_cap;                // LDS sort capacity of the select_decode variant in use; survivor keys per segment
  const uint32_t *counts;
  const uint64_t *cand;
  float *out_scores;     // [batch, n_levels*top_n]
  float *out_boxes;      // [batch, n_levels*top_n, NB]
  float *out_classes;    // [batch, n_levels*top_n]
  int32_t *out_indices;  // optional
  uint32_t *run_valid;   // optional [batch, n_levels]: emitted entries with score > 0 of every (image, level) list (nms sorted-run mode)
  int n_levels, batch, num_anchors, num_classes, top_n;
  float thresh;
  unsigned long long *trace;   // debug (odtk_debug_set_trace): 8 timestamps per workgroup, or null
};

// ---- key sources -------------------------------------------------------------------------
struct ListSource {   // the kSubLists compacted candidate sub-lists written by prefilter_scan_kernel
  const uint64_t *keys;              // sub-list s starts at keys + s * cap
  uint32_t cap;
  uint32_t start[kSubLists + 1];     // exclusive prefix of the (clamped) sub-list lengths: wave-uniform
  __device__ ListSource(const uint64_t *k, uint32_t n_flat) : keys(k), cap(n_flat) {   // ONE flat list of n_flat keys
#pragma unroll
    for (int s = 0; s <= kSubLists; ++s) start[s] = s == 0 ? 0 : n_flat;
  }
  __device__ ListSource(const uint64_t *k, const uint32_t *counts, uint32_t cap_) : keys(k), cap(cap_) {
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < kSubLists; ++s) {
      start[s] = acc;
      const uint32_t c = counts[s];
      acc += c < cap_ ? c : cap_;
    }
    start[kSubLists] = acc;
  }
  // one flat, fully occupied loop over all sub-lists (walking them one after the other would leave
  // most of the 1024 threads idle on the short lists and serialise 16 dependent count loads)
  template <typename F>
  __device__ __forceinline__ void for_each(F &&f) const {
    const uint32_t total = start[kSubLists];
    auto address = [&](uint32_t i) -> const uint64_t * { return address_of(i); };
    // The lists live in L2 and ONE workgroup walks them: its only source of memory-level parallelism
    // is independent loads per lane.  16 in flight per lane (128 KiB per workgroup), then 4, then 1.
    uint32_t i = threadIdx.x;
    i = batched<16>(i, total, address, f);
    i = batched<4>(i, total, address, f);
    for (; i < total; i += kSelThreads) f(*address(i));
  }
  __device__ __forceinline__ const uint64_t *address_of(uint32_t i) const {
    uint32_t sl = 0, base = 0;
#pragma unroll
    for (int q = 1; q < kSubLists; ++q)
      if (i >= start[q]) { sl = q; base = start[q]; }
    return keys + static_cast<uint64_t>(sl) * cap + (i - base);
  }
  // keys [lo, hi) of the flat order; every lane of the workgroup calls f(key, valid) the same number of times
  // (wave-level ballots inside f stay legal)
  template <int kThreads, typename F>
  __device__ __forceinline__ void for_range(uint32_t lo, uint32_t hi, F &&f) const {
    // the lists come out of L2 at ~1.5 us per dependent round trip: a slice of kSelSlice keys is ONE round (8 loads per lane)
    constexpr int kLoads = 8;
    for (uint32_t i0 = lo; i0 < hi; i0 += kLoads * kThreads) {
      uint64_t k[kLoads];
#pragma unroll
      for (int u = 0; u < kLoads; ++u) {
        const uint32_t i = i0 + u * kThreads + threadIdx.x;
        k[u] = i < hi ? *address_of(i) : 0;                 // 0 is not a key (an index never has all bits set)
      }
#pragma unroll
      for (int u = 0; u < kLoads; ++u) f(k[u], k[u] != 0);
    }
  }
  template <int kBatch, typename A, typename F>
  static __device__ __forceinline__ uint32_t batched(uint32_t i, uint32_t total, A &&address, F &&f) {
    for (; i + (kBatch - 1) * kSelThreads < total; i += kBatch * kSelThreads) {
      uint64_t k[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) k[u] = *address(i + u * kSelThreads);
#pragma unroll
      for (int u = 0; u < kBatch; ++u) f(k[u]);
    }
    return i;
  }
};
struct LdsSource {    // keys already gathered into LDS
  const uint64_t *keys;
  uint32_t count;
  template <typename F>
  __device__ __forceinline__ void for_each(F &&f) const {
    for (uint32_t i = threadIdx.x; i < count; i += kSelThreads) f(keys[i]);
  }
};
template <typename T, bool kLogits>
struct RawSource {    // the segment's raw head values (overflow path); walks memory order
  const void *image;  // first element of this image
  uint32_t n, channels, hw, channels_last;
  float thresh;
  const float *bias;  // per-channel head bias (logits only) or null
  template <typename F>
  __device__ __forceinline__ void for_each(F &&f) const {
    for (uint32_t r = threadIdx.x; r < n; r += kSelThreads) {
      float raw = load_raw<T>(image, r);
      if (kLogits && bias) raw += bias[channels_last ? r % channels : (r / hw) % channels];
      const float s = score_of<T, kLogits>(raw);
      if (s >= thresh) {
        uint32_t i = r;
        if (channels_last) { const uint32_t pix = r / channels, ch = r - pix * channels; i = ch * hw + pix; }
        f(make_key(s, i));
      }
    }
  }
};

// ---- block-wide helpers --------------------------------------------------------------------
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int lane_mask) {
  const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), lane_mask, kWave);
  const uint32_t hi = __shfl_xor(static_cast<uint32_t>(v >> 32), lane_mask, kWave);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// Bitonic sort, descending, of s_keys[0 .. 1024*E), 1024 threads, E keys per thread (element
// i = tid*E + e).  A compare-exchange with partner distance j needs
//   j <  E      : nothing but the thread's own registers,
//   j <  64*E   : one wave shuffle (partner lane = lane ^ j/E),
//   j >= 64*E   : another wave -> LDS + barrier.
// Of the 55 / 66 / 78 stages of a 1024 / 2048 / 4096-key network only 10 are of the last kind, so
// the keys live in registers and go through LDS only for those: ~20 barriers instead of 55-78
// (measured on MI355X: a 1024-key sort 13.6 us -> see DESIGN.md; every stage of the plain LDS
// version costs a full barrier round, ~0.25 us).
template <int E>
__device__ void bitonic_sort_desc_regs(uint64_t *s_keys) {
  constexpr uint32_t n = kSelThreads * E;
  const uint32_t tid = threadIdx.x;
  uint64_t v[E];
  auto load = [&] {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = s_keys[tid * E + e];
  };
  auto store = [&] {
#pragma unroll
    for (int e = 0; e < E; ++e) s_keys[tid * E + e] = v[e];
  };
  // all stages j = j_start .. 1 of phase k, for j_start < 64*E: registers + wave shuffles only
  auto reg_stages = [&](uint32_t k, uint32_t j_start) {
    for (uint32_t j = j_start; j >= static_cast<uint32_t>(E); j >>= 1) {   // partner in another lane
      const int lane_mask = static_cast<int>(j / E);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint64_t p = shfl_xor_u64(v[e], lane_mask);
        const uint32_t i = tid * E + e;
        const bool take_max = ((i & j) == 0) == ((i & k) == 0);   // lower element of a descending pair
        v[e] = take_max ? (v[e] > p ? v[e] : p) : (v[e] < p ? v[e] : p);
      }
      if (j == 1) return;                                         // E == 1: j ran down to 1 here
    }
#pragma unroll
    for (int j = E / 2; j > 0; j >>= 1) {                        // partner in the same thread
      if (static_cast<uint32_t>(j) > j_start) continue;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if ((e & j) == 0) {
          const uint32_t i = tid * E + e;
          const bool desc = (i & k) == 0;
          const uint64_t x = v[e], y = v[e | j];
          if (desc ? (x < y) : (x > y)) { v[e] = y; v[e | j] = x; }
        }
      }
    }
  };
  load();
  for (uint32_t k = 2; k <= 64u * E; k <<= 1) reg_stages(k, k >> 1);           // no LDS, no barrier
  for (uint32_t k = 128u * E; k <= n; k <<= 1) {
    store();
    __syncthreads();
    for (uint32_t j = k >> 1; j >= 64u * E; j >>= 1) {                          // cross-wave stages
      for (uint32_t t = tid; t < (n >> 1); t += kSelThreads) {
        const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const uint32_t hi = lo | j;
        const uint64_t x = s_keys[lo], y = s_keys[hi];
        const bool desc = (lo & k) == 0;
        if (desc ? (x < y) : (x > y)) { s_keys[lo] = y; s_keys[hi] = x; }
      }
      __syncthreads();
    }
    load();
    reg_stages(k, 32u * E);
  }
  store();
  __syncthreads();
}

// Sorts s_keys[0..n_valid) descending; entries up to the padded size are zeroed (sort last).
// The buffer must hold max(1024, pow2(n_valid)) <= kSortCapBig keys.
template <int kMaxKeys = 4096>
__device__ __forceinline__ void sort_keys_desc(uint64_t *s_keys, uint32_t n_valid) {
  uint32_t n_pad = kSelThreads;
  while (n_pad < n_valid) n_pad <<= 1;
  for (uint32_t i = n_valid + threadIdx.x; i < n_pad; i += kSelThreads) s_keys[i] = 0;
  __syncthreads();
  if (n_pad == kSelThreads) bitonic_sort_desc_regs<1>(s_keys);
  else if (n_pad == 2 * kSelThreads) bitonic_sort_desc_regs<2>(s_keys);
  else if (n_pad == 4 * kSelThreads || kMaxKeys <= 4 * kSelThreads) bitonic_sort_desc_regs<4>(s_keys);
  else if constexpr (kMaxKeys > 4 * kSelThreads) {           // only the top_n > 4096 variant carries the big networks
    if (n_pad == 8 * kSelThreads) bitonic_sort_desc_regs<8>(s_keys);
    else bitonic_sort_desc_regs<16>(s_keys);
  }
}

// Given a histogram in s_hist (kRadixBins bins, REVERSED: bin 0 = largest digit) finds the bin in which the running
// count (from the largest digit down) crosses `remaining`.  All threads return the same (bin, count above it, count
// inside it).  s_misc: [0..15] wave totals, [16..18] result.  Ends with a barrier; s_hist may be reused afterwards.
__device__ __forceinline__ void scan_boundary(const uint32_t *s_hist, uint32_t remaining, uint32_t *s_misc, uint32_t *rbin,
                                              uint32_t *above, uint32_t *in_bin) {
  // inclusive scan over kRadixBins bins, 2 per thread
  const uint32_t h0 = s_hist[2 * threadIdx.x], h1 = s_hist[2 * threadIdx.x + 1];
  const uint32_t inc = wave_inclusive_sum(h0 + h1);
  const int w = threadIdx.x >> 6;
  if (lane_id() == kWave - 1) s_misc[w] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int i = 0; i < w; ++i) woff += s_misc[i];
  const uint32_t excl = woff + inc - (h0 + h1);
  // the unique bin where the running count crosses `remaining`
  if (excl < remaining && remaining <= excl + h0) { s_misc[16] = 2 * threadIdx.x; s_misc[17] = excl; s_misc[18] = h0; }
  else if (excl + h0 < remaining && remaining <= excl + h0 + h1) { s_misc[16] = 2 * threadIdx.x + 1; s_misc[17] = excl + h0; s_misc[18] = h1; }
  __syncthreads();
  // (values read from LDS are VGPRs -- "divergent" to the compiler; callers steer loops with them, so pin them to SGPRs:
  // the descent loops then compile to scalar control flow instead of exec-masked waterfall loops)
  *rbin = __builtin_amdgcn_readfirstlane(s_misc[16]);
  *above = __builtin_amdgcn_readfirstlane(s_misc[17]);
  *in_bin = __builtin_amdgcn_readfirstlane(s_misc[18]);
  __syncthreads();
}

// MSD radix descent (11-bit digits, LDS histogram) on the bin that holds the `want`-th largest key.
// Returns a threshold T and *n_out = #{key >= T} with  want <= *n_out <= max_take : the descent
// stops as soon as everything above the boundary bin plus the bin itself fits `max_take`, so the
// caller sorts a few extra keys instead of paying for more passes.  Keys are unique; the source
// must hold at least `want` keys and max_take >= want.  s_misc: [0..15] wave totals, [16..18] result.
template <typename Source>
__device__ uint64_t radix_threshold(const Source &src, uint32_t want, uint32_t max_take, uint32_t *s_hist,
                                    uint32_t *s_misc, uint32_t *n_out) {
  uint64_t prefix = 0, pmask = 0;
  uint32_t remaining = want, taken_above = 0, in_bin = 0;
  int hi_bit = 64;
  while (hi_bit > 0) {
    const int bits = hi_bit >= kRadixBits ? kRadixBits : hi_bit;
    const int shift = hi_bit - bits;
    const uint32_t nb = 1u << bits;
    for (uint32_t i = threadIdx.x; i < kRadixBins; i += kSelThreads) s_hist[i] = 0;
    __syncthreads();
    // histogram, bins reversed so that an ascending scan walks keys from the largest digit down
    src.for_each([&](uint64_t key) {
      if ((key & pmask) == prefix) atomicAdd(&s_hist[(nb - 1) - static_cast<uint32_t>((key >> shift) & (nb - 1))], 1u);
    });
    __syncthreads();
    uint32_t rbin, above;
    scan_boundary(s_hist, remaining, s_misc, &rbin, &above, &in_bin);
    const uint64_t digit = (nb - 1) - rbin;
    prefix |= digit << shift;
    pmask |= static_cast<uint64_t>(nb - 1) << shift;
    remaining -= above;
    taken_above += above;
    hi_bit = shift;
    if (taken_above + in_bin <= max_take) break;   // at the last digit in_bin == remaining == 1
  }
  *n_out = taken_above + in_bin;
  return prefix;                                    // undecided low bits are 0 = start of the boundary bin
}

// ---- multi-workgroup narrowing (two launches in front of select_decode_kernel) ----------------------------
// A digit cuts the current key range [lo, hi] into 2048 equal bins (digit = (key - lo) >> sh) and histograms the keys
// inside it; the bin in which the running count, from the top, crosses top_n becomes the next range.
//   launch 0: first digit over [key(thresh, last index), key(1.0 | +inf, index 0)] -- every candidate lies in it; also
//             records the smallest and the largest key.
//   launch 1: FILTER -- every key >= the boundary bin's lower end goes to the segment's survivor list: the `taken` keys
//             above the bin, all wanted, and the bin's own `in_bin` keys of which select_decode keeps the `need` largest.
//             Where more keys than the survivor list holds share the boundary bin (saturated scores: all keys share
//             their score bits and differ only in the index bits) a SECOND DIGIT comes first, inside the same launch:
//             over the bin clipped to [min key, max key] (the clip makes the 2048 bins land on the bits that differ); the
//             workgroup that adds its histogram last finishes the segment (select_pass_kernel).
// Up to kSelParts workgroups of 256 threads per segment walk disjoint slices of the candidate lists (or of the raw
// scores when a sub-list overflowed); segments with <= sort-size candidates leave both launches at once.
struct SelState {
  uint64_t lo, hi;                      // current range, both ends inclusive
  uint32_t remaining, taken, in_bin, done;
};

__device__ __forceinline__ uint32_t sort_size_for(uint32_t top_n) {
  uint32_t sort_size = kSelThreads;
  while (sort_size < top_n) sort_size <<= 1;
  return sort_size;
}

__device__ __forceinline__ int range_shift(uint64_t lo, uint64_t hi) {   // (hi - lo) >> shift < 2048
  const uint64_t w = hi - lo;
  const int bl = w ? 64 - __clzll(static_cast<long long>(w)) : 0;
  return bl > kRadixBits ? bl - kRadixBits : 0;
}

// scan_boundary for a 256-thread workgroup: 8 consecutive bins per thread.  s_misc: [0..3] wave totals, [16..18] result.
__device__ __forceinline__ void scan_boundary_256(const uint32_t *s_hist, uint32_t remaining, uint32_t *s_misc, uint32_t *rbin,
                                                  uint32_t *above, uint32_t *in_bin) {
  constexpr int kPerThread = kRadixBins / kPassThreads;
  uint32_t h[kPerThread], sum = 0;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) { h[k] = s_hist[threadIdx.x * kPerThread + k]; sum += h[k]; }
  const uint32_t inc = wave_inclusive_sum(sum);
  const int w = threadIdx.x >> 6;
  if (lane_id() == kWave - 1) s_misc[w] = inc;
  __syncthreads();
  uint32_t run = inc - sum;
  for (int i = 0; i < w; ++i) run += s_misc[i];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    if (run < remaining && remaining <= run + h[k]) { s_misc[16] = threadIdx.x * kPerThread + k; s_misc[17] = run; s_misc[18] = h[k]; }
    run += h[k];
  }
  __syncthreads();
  // (values read from LDS are VGPRs -- "divergent" to the compiler; callers steer loops with them, so pin them to SGPRs:
  // the descent loops then compile to scalar control flow instead of exec-masked waterfall loops)
  *rbin = __builtin_amdgcn_readfirstlane(s_misc[16]);
  *above = __builtin_amdgcn_readfirstlane(s_misc[17]);
  *in_bin = __builtin_amdgcn_readfirstlane(s_misc[18]);
  __syncthreads();
}

// Folds one histogram pass into the state (block-wide, uniform result).  [kmin, kmax]: all keys of the segment.
// kFresh: the histogram was completed by OTHER workgroups of this launch (ticket): read it past this CU's vector cache.
template <bool kFresh = false>
__device__ __forceinline__ void advance_state(SelState &st, const uint32_t *g_hist, uint64_t kmin, uint64_t kmax, uint32_t sort_cap,
                                              uint32_t *s_hist, uint32_t *s_misc) {
  const int sh = range_shift(st.lo, st.hi);
  for (uint32_t i = threadIdx.x; i < kRadixBins; i += kPassThreads)
    s_hist[i] = kFresh ? __hip_atomic_load(g_hist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g_hist[i];
  __syncthreads();
  uint32_t rbin, above, in_bin;
  scan_boundary_256(s_hist, st.remaining, s_misc, &rbin, &above, &in_bin);
  const uint64_t digit = (kRadixBins - 1) - rbin;
  uint64_t lo = st.lo + (digit << sh);
  const uint64_t span = sh ? ((1ull << sh) - 1ull) : 0ull;
  uint64_t hi = lo > ~0ull - span ? ~0ull : lo + span;
  hi = hi < st.hi ? hi : st.hi;
  lo = lo > kmin ? lo : kmin;                               // no key lies outside [kmin, kmax]
  hi = hi < kmax ? hi : kmax;
  st.lo = lo;
  st.hi = hi;
  st.remaining -= above;
  st.taken += above;
  st.in_bin = in_bin;
  if ((in_bin <= kRankCap && st.taken + in_bin <= sort_cap) || lo >= hi) st.done = 1;
}

// radix_threshold on a KNOWN key range [lo, hi] (both inclusive), 1024-thread workgroups: every pass cuts the range into
// 2048 equal bins instead of taking the next 11 key bits, so the first pass already lands on the bits in which the keys
// differ (fp32 scores of one image share their exponent bits: an 11-bit MSD digit separates almost nothing).  The source
// must hold at least `want` keys inside the range.  Returns T with  want <= #{key in [T, hi]} = *n_out <= max_take  (or
// the exact `want`-th key when ties in the digit cannot be split further).
template <typename Source>
__device__ uint64_t range_threshold(const Source &src, uint32_t want, uint32_t max_take, uint64_t lo, uint64_t hi, uint32_t *s_hist,
                                    uint32_t *s_misc, uint32_t *n_out) {
  uint32_t remaining = want, taken_above = 0, in_bin = 0;
  for (;;) {
    const int sh = range_shift(lo, hi);
    for (uint32_t i = threadIdx.x; i < kRadixBins; i += kSelThreads) s_hist[i] = 0;
    __syncthreads();
    src.for_each([&](uint64_t key) {
      if (key >= lo && key <= hi) {
        uint32_t digit = static_cast<uint32_t>((key - lo) >> sh);
        digit = digit > kRadixBins - 1 ? kRadixBins - 1 : digit;
        atomicAdd(&s_hist[(kRadixBins - 1) - digit], 1u);
      }
    });
    __syncthreads();
    uint32_t rbin, above;
    scan_boundary(s_hist, remaining, s_misc, &rbin, &above, &in_bin);
    const uint64_t digit = (kRadixBins - 1) - rbin;
    const uint64_t span = sh ? ((1ull << sh) - 1ull) : 0ull;
    const uint64_t nlo = lo + (digit << sh);
    uint64_t nhi = nlo > ~0ull - span ? ~0ull : nlo + span;
    nhi = nhi < hi ? nhi : hi;
    lo = nlo;
    hi = nhi;
    remaining -= above;
    taken_above += above;
    if (taken_above + in_bin <= max_take || sh == 0) break;
  }
  *n_out = taken_above + in_bin;
  return lo;
}

// The raw head values of elements [lo, hi) (memory order) of one image, as keys: the overflow path of the passes.
template <typename T, bool kLogits>
struct RawSlice {
  const void *image;
  uint32_t n, channels, hw, channels_last;
  float thresh;
  const float *bias;
  template <int kThreads, typename F>
  __device__ __forceinline__ void for_range(uint32_t lo, uint32_t hi, F &&f) const {
    for (uint32_t r0 = lo; r0 < hi; r0 += 8 * kThreads) {
      float raw[8];
      bool ok[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t r = r0 + u * kThreads + threadIdx.x;
        ok[u] = r < hi;
        raw[u] = ok[u] ? load_raw<T>(image, r) : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t r = r0 + u * kThreads + threadIdx.x;
        float x = raw[u];
        bool take = ok[u];
        uint64_t key = 0;
        if (take) {
          if (kLogits && bias) x += bias[channels_last ? r % channels : (r / hw) % channels];
          const float s = score_of<T, kLogits>(x);
          take = s >= thresh;
          if (take) {
            uint32_t i = r;
            if (channels_last) { const uint32_t pix = r / channels, ch = r - pix * channels; i = ch * hw + pix; }
            key = make_key(s, i);
          }
        }
        f(key, take);
      }
    }
  }
};

// PASS 0: the first histogram (+ smallest / largest key).
// PASS 1: what follows it, in ONE launch (round 3; there were two, the first of them launched and skipped in the normal case):
//   * every workgroup derives the state after pass 0 from the segment's histogram (8 KiB out of L2);
//   * boundary bin rankable, or everything at or above it fits the survivor list  -> FILTER now: this workgroup's slice of
//     the keys >= the bin's lower end goes to the survivor list (select_decode ranks a bin of <= kRankCap keys by brute
//     force and radix-selects a larger one in LDS);
//   * otherwise (saturated scores, plateaus: > surv_cap keys share the boundary bin)  -> SECOND DIGIT: histogram of the
//     slice over the bin, clipped to [min key, max key]; the workgroup that arrives LAST at the segment's ticket counter
//     (no grid barrier, nobody waits) folds that histogram into the state and filters the WHOLE segment alone -- slower
//     than 64 workgroups, but it is the rare route and costs the common one no launch.
template <typename T, bool kLogits, int PASS>
__global__ __launch_bounds__(kPassThreads) void select_pass_kernel(const DecodeArgs a) {
  __shared__ __attribute__((aligned(8))) uint32_t s_hist[kRadixBins];
  __shared__ uint32_t s_misc[32];
  __shared__ unsigned long long s_range[2];

  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.part_begin[i]) l = i;
  const uint32_t P = a.parts[l];
  const uint32_t j = blockIdx.x - a.part_begin[l];
  const uint32_t b = j / P, part = j - b * P;
  const int seg = l * a.batch + static_cast<int>(b);
  const DecodeLevel &L = a.lv[l];
  const uint32_t *sub_counts = a.counts + static_cast<size_t>(seg) * kSubLists;
  uint32_t count = 0;
  bool complete = true;
#pragma unroll
  for (int s = 0; s < kSubLists; ++s) {
    const uint32_t c = sub_counts[s];
    count += c;
    complete = complete && c <= L.cap;
  }
  if (count <= sort_size_for(a.top_n)) return;              // select_decode sorts these directly (block-uniform exit)
  SelSeg &S = a.sel[seg];
  // debug trace (odtk_debug_set_trace): 5 timestamps of part 0 per (pass, segment), behind the select_decode / nms slots
  auto stamp = [&](int k) {
    if (a.trace && part == 0 && threadIdx.x == 0) a.trace[1024 + (PASS * 64 + seg) * 8 + k] = wall_clock64();
  };
  stamp(0);

  // ---- this workgroup's slice of the segment ----
  const uint32_t hw = static_cast<uint32_t>(L.height) * L.width;
  const uint32_t channels = static_cast<uint32_t>(a.num_anchors) * a.num_classes;
  const ListSource lists(a.cand + L.cand_off + static_cast<uint64_t>(b) * kSubLists * L.cap, sub_counts, L.cap);
  const uint32_t total = complete ? lists.start[kSubLists] : L.n;
  uint32_t active = complete ? (total + kSelSlice - 1) / kSelSlice : P;
  if (active > P) active = P;
  const uint32_t chunk = ((total + active - 1) / active + kPassThreads - 1) / kPassThreads * kPassThreads;
  const uint32_t n_live = (total + chunk - 1) / chunk;       // workgroups of this segment that own a non-empty slice
  const uint32_t lo = part * chunk;
  const uint32_t hi = lo + chunk < total ? lo + chunk : total;
  if (part >= active || lo >= hi) return;
  const typename T::storage *cls_image = static_cast<const typename T::storage *>(L.cls) + static_cast<uint64_t>(b) * L.n;
  const RawSlice<T, kLogits> raw{cls_image, L.n, channels, hw, L.channels_last, a.thresh, L.cls_bias};
  auto walk = [&](uint32_t w_lo, uint32_t w_hi, auto &&fn) {
    if (complete) lists.template for_range<kPassThreads>(w_lo, w_hi, fn);
    else raw.template for_range<kPassThreads>(w_lo, w_hi, fn);
  };
  const int lane = lane_id();

  // histogram of the keys inside [r_lo, r_hi] (2048 equal bins, reversed) into g_hist; pass 0 also records min / max
  auto histogram = [&](uint64_t r_lo, uint64_t r_hi, bool clip, uint32_t *g_hist) {
    for (uint32_t i = threadIdx.x; i < kRadixBins; i += kPassThreads) s_hist[i] = 0;
    if (threadIdx.x < 2) s_range[threadIdx.x] = 0;
    __syncthreads();
    const int sh = range_shift(r_lo, r_hi);
    uint64_t my_max = 0, my_min_inv = 0;
    walk(lo, hi, [&](uint64_t key, bool valid) {
      if (clip) valid = valid && key >= r_lo && key <= r_hi;
      const uint64_t m = __ballot(valid);
      if (!m) return;                                       // wave-uniform
      uint32_t digit = static_cast<uint32_t>((key - r_lo) >> sh);
      digit = digit > kRadixBins - 1 ? kRadixBins - 1 : digit;
      const uint32_t bin = (kRadixBins - 1) - digit;
      // a wave whose lanes all hit ONE bin (saturated inputs: every key) adds once -- 64 LDS atomics on one word
      // would serialise
      const int leader = __ffsll(static_cast<unsigned long long>(m)) - 1;
      const uint32_t bin0 = __shfl(bin, leader, kWave);
      const uint64_t same = __ballot(valid && bin == bin0);
      if (same == m) { if (lane == leader) atomicAdd(&s_hist[bin0], static_cast<uint32_t>(__popcll(m))); }
      else if (valid) atomicAdd(&s_hist[bin], 1u);
      if (PASS == 0 && valid) {
        my_max = key > my_max ? key : my_max;
        my_min_inv = ~key > my_min_inv ? ~key : my_min_inv;
      }
    });
    if (PASS == 0) {
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o1 = shfl_xor_u64(my_max, d), o2 = shfl_xor_u64(my_min_inv, d);
        my_max = o1 > my_max ? o1 : my_max;
        my_min_inv = o2 > my_min_inv ? o2 : my_min_inv;
      }
      if (lane == 0) { atomicMax(&s_range[0], my_max); atomicMax(&s_range[1], my_min_inv); }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kRadixBins; i += kPassThreads) {
      const uint32_t h = s_hist[i];
      if (h) atomicAdd(&g_hist[i], h);
    }
    if (PASS == 0 && threadIdx.x == 0) { atomicMax(&S.kmax, s_range[0]); atomicMax(&S.kmin_inv, s_range[1]); }
  };

  // pass 0 range: every candidate has score >= thresh; sigmoid outputs never exceed 1
  SelState st;
  st.lo = make_key(a.thresh, 0xffffffffu);
  st.hi = kLogits ? make_key(1.0f, 0u) : ~0ull;
  st.remaining = static_cast<uint32_t>(a.top_n);
  st.taken = st.in_bin = st.done = 0;
  if (PASS == 0) {
    stamp(1);
    histogram(st.lo, st.hi, false, S.hist[0]);
    stamp(4);
    return;
  }

  // ---- PASS 1 ----
  advance_state(st, S.hist[0], ~S.kmin_inv, S.kmax, a.sort_cap, s_hist, s_misc);
  uint64_t *surv = a.surv + static_cast<uint64_t>(seg) * a.surv_cap;
  constexpr uint32_t kStageKeys = kRadixBins / 2;           // s_hist reinterpreted as 64-bit keys
  uint64_t *s_stage = reinterpret_cast<uint64_t *>(s_hist);
  // every key >= T64 of [w_lo, w_hi) goes to the survivor list: staged in LDS (the histogram's 8 KiB = 1024 keys, idle now),
  // ONE returning global atomic per workgroup and stage-full; a returning atomic per wave and round cost 1-2 us each
  auto filter = [&](uint64_t T64, uint32_t w_lo, uint32_t w_hi) {
    if (threadIdx.x == 0) s_misc[24] = 0;
    __syncthreads();
    auto flush = [&]() {                                     // block-uniform call sites
      __syncthreads();
      const uint32_t staged = s_misc[24] < kStageKeys ? s_misc[24] : kStageKeys;
      if (staged) {
        if (threadIdx.x == 0) s_misc[25] = atomicAdd(&S.surv_count, staged);
        __syncthreads();
        const uint32_t g0 = s_misc[25];
        for (uint32_t i = threadIdx.x; i < staged; i += kPassThreads)
          if (g0 + i < a.surv_cap) surv[g0 + i] = s_stage[i];
      }
      __syncthreads();
      if (threadIdx.x == 0) s_misc[24] = 0;
      __syncthreads();
    };
    // slabs of one load round (8 keys per lane): the stage (1024 keys) cannot overflow inside a slab of 2048 keys twice
    for (uint32_t s_lo = w_lo; s_lo < w_hi; s_lo += kSelSlice) {
      const uint32_t s_hi = s_lo + kSelSlice < w_hi ? s_lo + kSelSlice : w_hi;
      walk(s_lo, s_hi, [&](uint64_t key, bool valid) {
        const bool take = valid && key >= T64;
        const uint64_t m = __ballot(take);
        if (!m) return;
        const int leader = __ffsll(static_cast<unsigned long long>(m)) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_misc[24], static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, leader, kWave);
        if (take) {
          const uint32_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < kStageKeys) s_stage[pos] = key;
          else { const uint32_t g = atomicAdd(&S.surv_count, 1u); if (g < a.surv_cap) surv[g] = key; }   // stage full (rare)
        }
      });
      if (s_hi < w_hi) flush();                              // (only the last workgroup of a second digit walks more than one slab)
    }
    flush();
  };
  auto publish = [&](const SelState &f, bool fits) {
    S.T = f.lo; S.bin_hi = f.hi; S.expected = f.taken + f.in_bin; S.need = f.remaining; S.filtered = fits ? 1u : 0u;
    S.n_above = f.taken; S.n_bin = f.in_bin;
  };
  stamp(1);
  if (st.done || st.taken + st.in_bin <= a.surv_cap) {
    // the common routes: filter this slice now
    if (part == 0 && threadIdx.x == 0) publish(st, true);
    stamp(2);
    filter(st.lo, lo, hi);
    stamp(4);
    return;
  }
  // second digit: this slice's histogram over the boundary bin, then the ticket
  histogram(st.lo, st.hi, true, S.hist[1]);
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_misc[26] = atomicAdd(&S.arrived, 1u);
  __syncthreads();
  if (s_misc[26] != n_live - 1) return;                      // (block-uniform) somebody else is last
  __threadfence();
  advance_state<true>(st, S.hist[1], st.lo, st.hi, a.sort_cap, s_hist, s_misc);
  const bool fits = st.taken + st.in_bin <= a.surv_cap;
  if (threadIdx.x == 0) publish(st, fits);
  if (fits) filter(st.lo, 0, total);                         // (adversarial key sets that two digits cannot split: select_decode walks the source itself)
  stamp(4);
}

// ---- the kernel ------------------------------------------------------------------------------
// NB: box parameters (4 axis-aligned, 6 rotated); T: element type of BOTH head tensors;
// kLogits: cls holds logits (sigmoid fused, see prefilter.hpp score_of).
// CAP: keys the LDS sort buffer holds -- kSortCap (static LDS) for top_n <= 4096, kSortCapBig (dynamic LDS, the launch
// passes CAP * 8 bytes) beyond.
template <int NB, typename T, bool kLogits, int CAP = kSortCap>
__global__ __launch_bounds__(kSelThreads) void select_decode_kernel(const DecodeArgs a) {
  __shared__ uint64_t s_keys_static[CAP <= kSortCap ? CAP : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char s_keys_dynamic[];
  uint64_t *s_keys = CAP <= kSortCap ? s_keys_static : reinterpret_cast<uint64_t *>(s_keys_dynamic);
  __shared__ uint32_t s_hist[kRadixBins];
  __shared__ uint32_t s_misc[32];

  const int seg = blockIdx.x;
  const int l = seg / a.batch;
  const int b = seg - l * a.batch;
  const DecodeLevel &L = a.lv[l];
  // exact survivor count = sum over the sub-lists; the lists are complete iff none overflowed
  const uint32_t *sub_counts = a.counts + static_cast<size_t>(seg) * kSubLists;
  uint32_t count = 0;
  bool complete = true;
#pragma unroll
  for (int s = 0; s < kSubLists; ++s) {
    const uint32_t c = sub_counts[s];
    count += c;
    complete = complete && c <= L.cap;
  }
  const uint32_t top_n = a.top_n;
  const uint32_t k_out = count < top_n ? count : top_n;
  const int H = L.height, W = L.width, A = a.num_anchors, C = a.num_classes;
  const uint32_t hw = static_cast<uint32_t>(H) * W;
  const uint32_t channels = static_cast<uint32_t>(A) * C;
  const typename T::storage *cls_image = static_cast<const typename T::storage *>(L.cls) + static_cast<uint64_t>(b) * L.n;

  auto stamp = [&](int k) { if (a.trace && threadIdx.x == 0) a.trace[blockIdx.x * 8 + k] = wall_clock64(); };
  stamp(0);
  if (threadIdx.x == 0) s_misc[22] = 0;                                  // positive scores emitted (run_valid)
  const ListSource lists(a.cand + L.cand_off + static_cast<uint64_t>(b) * kSubLists * L.cap, sub_counts, L.cap);
  uint32_t n_sort;   // number of valid keys placed in s_keys

  // more candidates than one sort holds: the multi-workgroup passes (select_pass_kernel) have normally left the
  // keys at or above the boundary bin -- top_n plus a few -- in the segment's survivor list
  const SelSeg *S = a.sel ? a.sel + seg : nullptr;
  const bool narrowed = S && count > sort_size_for(top_n) && S->filtered != 0;
  if (narrowed) {
    const uint32_t n_surv = S->expected;                               // <= kSurvCap, >= top_n
    const ListSource surv(a.surv + static_cast<uint64_t>(seg) * a.surv_cap, n_surv);
    const uint32_t n_hi = S->n_above, in_bin = S->n_bin, need = S->need;   // n_hi + in_bin == n_surv, n_hi + need == top_n
    if (in_bin <= kRankCap && n_surv <= static_cast<uint32_t>(CAP) && top_n + in_bin <= static_cast<uint32_t>(CAP)) {
      // the normal route: keys above the boundary bin go to the front of the sort buffer, the bin's own keys to its
      // back; then every bin key counts the bin keys larger than itself (keys are unique: the counts are the ranks) and
      // the `need` best land, already in order, behind the others -- exactly top_n keys, no further narrowing
      const uint64_t bin_hi = S->bin_hi;
      if (threadIdx.x == 0) { s_misc[20] = 0; s_misc[21] = 0; }
      __syncthreads();
      for (uint32_t i0 = 0; i0 < n_surv; i0 += kSelThreads) {            // (block-uniform trip count: ballots are legal)
        const uint32_t i = i0 + threadIdx.x;
        const uint64_t key = i < n_surv ? surv.keys[i] : 0;
        const bool above = key > bin_hi, inside = key != 0 && !above;
        const uint64_t m_a = __ballot(above), m_i = __ballot(inside);
        const int lane = lane_id();
        uint32_t base_a = 0, base_i = 0;
        if (lane == 0) {
          if (m_a) base_a = atomicAdd(&s_misc[20], static_cast<uint32_t>(__popcll(m_a)));
          if (m_i) base_i = atomicAdd(&s_misc[21], static_cast<uint32_t>(__popcll(m_i)));
        }
        base_a = __shfl(base_a, 0, kWave);
        base_i = __shfl(base_i, 0, kWave);
        const uint64_t lt = (1ull << lane) - 1ull;
        if (above) s_keys[base_a + __popcll(m_a & lt)] = key;
        if (inside) s_keys[CAP - 1 - (base_i + __popcll(m_i & lt))] = key;
      }
      __syncthreads();
      if (threadIdx.x < in_bin) {
        const uint64_t mine = s_keys[CAP - 1 - threadIdx.x];
        uint32_t rank = 0;
        for (uint32_t q = 0; q < in_bin; ++q) rank += s_keys[CAP - 1 - q] > mine ? 1u : 0u;   // same address in every lane: broadcast
        if (rank < need) s_keys[n_hi + rank] = mine;
      }
      n_sort = n_hi + need;
    } else {
      uint64_t T64 = 0;
      n_sort = n_surv;
      if (n_surv > CAP) T64 = radix_threshold(surv, top_n, CAP, s_hist, s_misc, &n_sort);   // tie-heavy inputs only
      if (threadIdx.x == 0) s_misc[20] = 0;
      __syncthreads();
      surv.template for_range<kSelThreads>(0, n_surv, [&](uint64_t key, bool valid) {
        const bool take = valid && key >= T64;
        const uint32_t slot = wave_append_slot(&s_misc[20], take);
        if (take && slot < CAP) s_keys[slot] = key;
      });
    }
  } else if (count <= CAP && complete) {
    if (threadIdx.x == 0) s_misc[20] = 0;
    __syncthreads();
    lists.template for_range<kSelThreads>(0, lists.start[kSubLists], [&](uint64_t key, bool valid) {   // order is irrelevant
      const uint32_t slot = wave_append_slot(&s_misc[20], valid);     // one LDS atomic per wave: 4096 on one word cost ~30 us
      if (valid) s_keys[slot] = key;
    });
    n_sort = count;
  } else {
    // (reached only when the passes declined: > kSurvCap keys share 22 leading key bits with the top_n-th)
    const RawSource<T, kLogits> raw{cls_image, L.n, channels, hw, L.channels_last, a.thresh, L.cls_bias};
    uint64_t T64 = 0;
    n_sort = count;                // count <= top_n: everything is wanted (overflow path only)
    if (count > top_n)
      T64 = complete ? radix_threshold(lists, top_n, CAP, s_hist, s_misc, &n_sort)
                     : radix_threshold(raw, top_n, CAP, s_hist, s_misc, &n_sort);
    if (threadIdx.x == 0) s_misc[20] = 0;
    __syncthreads();
    if (complete) {
      lists.template for_range<kSelThreads>(0, lists.start[kSubLists], [&](uint64_t key, bool valid) {
        const bool take = valid && key >= T64;
        const uint32_t slot = wave_append_slot(&s_misc[20], take);
        if (take && slot < CAP) s_keys[slot] = key;
      });
    } else {
      raw.for_each([&](uint64_t key) {
        if (key >= T64) { const uint32_t p = atomicAdd(&s_misc[20], 1u); if (p < CAP) s_keys[p] = key; }
      });
    }
  }
  stamp(1);
  __syncthreads();
  // Second stage, in LDS.  Sorting is the expensive part (measured: 1024 keys 7 us, 4096 keys 24 us)
  // while a radix pass over keys that are already LDS-resident costs ~3 us, so narrow the buffer down
  // to the smallest sortable size that still holds top_n (1024 for the default 1000) first.
  const uint32_t sort_size = sort_size_for(top_n);
  if (n_sort > sort_size) {
    uint32_t n_keep = n_sort;
    const LdsSource in_lds{s_keys, n_sort};
    const uint64_t T2 = radix_threshold(in_lds, top_n, sort_size, s_hist, s_misc, &n_keep);
    // in-place compaction: every lane reads its keys (<= 4), barrier, survivors go to the front
    uint64_t mine[CAP / kSelThreads];
#pragma unroll
    for (int u = 0; u < CAP / kSelThreads; ++u) {
      const uint32_t i = u * kSelThreads + threadIdx.x;
      mine[u] = i < n_sort ? s_keys[i] : 0;
    }
    if (threadIdx.x == 0) s_misc[20] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CAP / kSelThreads; ++u) {
      const bool keep = mine[u] != 0 && mine[u] >= T2;
      const uint32_t slot = wave_append_slot(&s_misc[20], keep);
      if (keep) s_keys[slot] = mine[u];
    }
    n_sort = n_keep;
    __syncthreads();
  }
  stamp(2);
  if (a.trace && threadIdx.x == 0) { a.trace[blockIdx.x * 8 + 5] = count; a.trace[blockIdx.x * 8 + 6] = n_sort; a.trace[blockIdx.x * 8 + 7] = complete; }
  sort_keys_desc<CAP>(s_keys, n_sort);   // the first k_out are the answer
  stamp(3);

  // ---- decode + write this segment's slice of the concatenated outputs ----
  const float stride = L.stride;
  const float lim_x = static_cast<float>(W) * stride - 1.0f;   // box.py:106  M = size*stride - 1
  const float lim_y = static_cast<float>(H) * stride - 1.0f;
  const typename T::storage *box_image = static_cast<const typename T::storage *>(L.box) + static_cast<uint64_t>(b) * A * NB * hw;
  const uint64_t out_row = static_cast<uint64_t>(b) * a.n_levels * top_n + static_cast<uint64_t>(l) * top_n;

  for (uint32_t t = threadIdx.x; t < top_n; t += kSelThreads) {
    float score = 0.0f, cls = 0.0f;
    float bx[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bx[k] = 0.0f;
    int32_t index = -1;
    if (t < k_out) {
      const uint64_t key = s_keys[t];
      const uint32_t i = key_index(key);
      index = static_cast<int32_t>(i);
      const uint32_t pix = i % hw;
      const uint32_t x = pix % W, y = pix / W;
      const uint32_t c = (i / hw) % C;
      const uint32_t an = i / (hw * C);
      cls = static_cast<float>(c);
      if (!kLogits && sizeof(typename T::storage) == 4)
        // the key canonicalises -0.0 to +0.0 for ordering; emit the stored value itself
        score = load_raw<T>(cls_image, memory_offset(i, channels, hw, L.channels_last));
      else
        score = key_score(key);
      float d[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const uint64_t off = L.channels_last ? static_cast<uint64_t>(pix) * (A * NB) + an * NB + k
                                             : (static_cast<uint64_t>(an) * NB + k) * hw + pix;
        d[k] = load_raw<T>(box_image, off);
        if (L.box_bias) d[k] += L.box_bias[an * NB + k];                  // head bias folded in (fp32 add)
      }
      // box.py:302  grid = [x, y, x, y] * stride + anchors[a]
      const float *anc = L.anchors + 4 * an;
      const float fx = static_cast<float>(x) * stride, fy = static_cast<float>(y) * stride;
      const float ax1 = fx + anc[0], ay1 = fy + anc[1], ax2 = fx + anc[2], ay2 = fy + anc[3];
      // box.py:100-103
      const float w = ax2 - ax1 + 1.0f, h = ay2 - ay1 + 1.0f;
      const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
      const float pcx = d[0] * w + cx, pcy = d[1] * h + cy;
      const float pw = exp_cr(d[2]) * w, ph = exp_cr(d[3]) * h;
      // box.py:108-111
      bx[0] = clamp_like_torch(pcx - 0.5f * pw, lim_x);
      bx[1] = clamp_like_torch(pcy - 0.5f * ph, lim_y);
      bx[2] = clamp_like_torch(pcx + 0.5f * pw - 1.0f, lim_x);
      bx[3] = clamp_like_torch(pcy + 0.5f * ph - 1.0f, lim_y);
      if constexpr (NB == 6) { bx[4] = d[4]; bx[5] = d[5]; }   // sin, cos pass through (decode_rotate.cu:152-162)
    }
    if (a.run_valid) {                                                   // (block-uniform; the list is sorted: positives are a prefix)
      const uint64_t positive = __ballot(score > 0.0f);
      if (positive && lane_id() == 0) atomicAdd(&s_misc[22], static_cast<uint32_t>(__popcll(positive)));
    }
    a.out_scores[out_row + t] = score;
    a.out_classes[out_row + t] = cls;
#pragma unroll
    for (int k = 0; k < NB; ++k) a.out_boxes[(out_row + t) * NB + k] = bx[k];
    if (a.out_indices) a.out_indices[out_row + t] = index;
  }
  if (a.run_valid) {
    __syncthreads();
    if (threadIdx.x == 0) a.run_valid[static_cast<size_t>(b) * a.n_levels + l] = s_misc[22];
  }
  stamp(4);
}

}  // namespace odtk

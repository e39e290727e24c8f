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
// IoU arithmetic is written in box.py's operation order (-ffp-contract=off); the rotated IoU follows
// csrc/cuda/nms_iou.cu:114-248 (rotated_iou.hpp).
//
// Why this shape.  The reference (and a straightforward port) lets every kept box "push"
// suppression onto all later boxes: count x kept IoU evaluations and one barrier per kept box, even
// though only the first `ndetections` survivors are emitted.  Here the boxes are consumed lazily, in
// score order, and each candidate "pulls" against the boxes kept so far:
//   round    : the next (up to) 1024 best candidates, in order.
//              generic input : radix-selected out of the LDS-resident key list and sorted (bitonic);
//              sorted runs   : (odtk_detect) the input is decode_levels' output, per level a list already in NMS
//                              order: the round is a prefix of every run, found by probing 1024 / n_runs slots per
//                              run, and its order comes from RANKS -- a key's rank is its own slot plus, per other
//                              run, a binary search -- not from a sort (round 3; the sort was 7 of the kernel's 29 us).
//   chunk    : 64 candidates (lane <-> candidate).
//              axis-aligned  : (1) all 16 waves test the SAME 64 candidates, each against its own 1/16 slice of the
//                              kept list, and compute the chunk's pairwise suppression rows; (2) wave 0 resolves the
//                              64 in rank order with scalar bit arithmetic.
//              rotated       : the polygon clip costs ~1000 wave instructions, so pairs are never evaluated where a
//                              lane happens to sit: every (kept box | earlier candidate, candidate) pair that shares
//                              a class and survives the cheap distance reject is APPENDED to a work queue in LDS, and
//                              the queue is drained densely, one pair per thread -- one clip call site in the kernel,
//                              no idle lanes beside a busy one, pull first and suppression rows only among the
//                              candidates the pull left alive.
//   filter   : when a round's yield says that every remaining candidate will have to be examined anyway (heavy
//              suppression: 2236 candidates for 9 kept boxes took eleven rounds), all of them are tested against the
//              whole kept list at once (1024 at a time, in parallel) and only the survivors stay listed.
#pragma once

#include "common.hpp"
#include "rotated_iou.hpp"
#include "select_decode.hpp"   // range_threshold, sort_keys_desc
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kNmsThreads = 1024;
constexpr int kNmsRound = 1024;        // keys selected + ordered per round (one per thread)
constexpr int kNmsChunk = 64;          // candidates resolved per chunk: one wave-width
constexpr int kPairQueue = kRadixBins; // rotated: (box, box) pairs per queue slab -- the queue lives in the histogram's 8 KiB
constexpr int kNmsMisc = 160;          // words of s_misc
constexpr uint32_t kNmsFlagChunks = 0x80000000u;   // internal (NmsArgs::flags): axis-aligned rounds through the chunk loop of rounds 3-5 (A/B)

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
                           // sets it (together with run_valid); the stand-alone nms entry points (arbitrary input) leave it 0.
  const uint32_t *run_valid;   // [batch, count / run_len]: entries with a positive score at the head of every run (select_decode)
  int ndet;
  float thresh;
  uint32_t flags;
  unsigned long long *trace;   // debug (odtk_debug_set_trace): 8 timestamps per workgroup, or null
  // rotated boxes, three launches (below): the first round in order, its pairwise suppression matrix, the resolve
  float *first_box;            // [batch, m_max, 6]  boxes of the first round's candidates, in NMS order
  float *first_cls;            // [batch, m_max]
  uint32_t *first_n;           // [batch]            how many of them the matrix covers
  unsigned long long *first_keys;   // [batch, kNmsRound] the whole first round's keys, in order (stage 1 -> stage 2)
  uint32_t *first_state;       // [batch, 16]        its size, and where the selection stood after it (run cursors | lower key bound)
  unsigned long long *sup;     // [batch, m_max, m_max / 64]  bit (j % 64) of word j / 64 of row i: candidate i suppresses candidate j (i < j)
  uint32_t m_max;              // multiple of 64, <= kNmsRound
  uint32_t m_first;            // two-step speculation: the first matrix launch covers candidates < m_first (multiple of 64) ...
  uint32_t step;               // ... stage 2 runs as step 1 (resolve those; done[img] = did it suffice) and step 2 (the rest); 0: one step
  uint32_t *done;              // [batch]
};

// Rotated boxes: the polygon clip is ~1000 wave instructions, and ONE workgroup per image -- one CU out of 256 -- cannot
// evaluate the thousands of pairs an image needs in less than ~100 us (measured: 3.5 us per 1024 clips, VALU-bound).  The
// pairs of the FIRST round are therefore evaluated by the whole chip:
//   stage 1  nms_kernel<6, ., 1>         one workgroup per image: the first round (up to m_max candidates) in NMS order
//                                        -> first_box / first_cls / first_n
//   matrix   rotated_sup_matrix_kernel   one workgroup per 64 x 64 tile of (i < j) pairs and image: class test, distance
//                                        reject, clip -> one 64-bit suppression word per (row, column block)
//   stage 2  nms_kernel<6, ., 2>         one workgroup per image: the same first round again (deterministic), resolved
//                                        from the matrix with bit arithmetic only -- a kept candidate ORs its row into the
//                                        dead bits of the columns behind it; anything beyond the matrix (more than m_max
//                                        candidates examined) continues with the in-workgroup pair queue.
// The matrix is speculative -- all pairs of the first m candidates, not only those a lazy pull would test -- which is what
// makes it parallel; m_max = 8 x detections_per_im bounds the waste.
struct SupArgs {
  const float *first_box;
  const float *first_cls;
  const uint32_t *first_n;
  unsigned long long *sup;
  uint32_t m_max;              // row stride of `sup` = m_max / 64 words
  uint32_t m_launch;           // candidates this launch covers (its grid holds the tiles of m_launch / 64 blocks)
  uint32_t m_done;             // tiles whose column block lies below m_done were written by an earlier launch
  const uint32_t *done;        // [batch] or null: images that need no more of the matrix
  float thresh;
  uint32_t flags;
};

constexpr int kSupThreads = 256;
constexpr uint32_t kSupRows = 4;       // rows of a 64 x 64 tile per workgroup: at most ONE clip per thread (16 rows: up to four, one
                                       // after the other, ~4 us each -- the densest slice set the launch's 18-21 us in round 3)

// One workgroup per kSupRows x 64 slice of a 64 x 64 tile of (i < j) candidate pairs and image (one clip per thread at most: a
// lone wave needs ~3..5 us per clip, so the work has to be wide, not deep).  A wave that clipped a whole row whenever ONE of its
// 64 columns needed it paid ~46 000 wave-clips at m = 800 (44 us, VALU-bound) for pairs of which a quarter needed the clip;
// so, as in the NMS kernel itself: enumerate the tile's pairs (class, distance reject), append the ones that need the
// polygon clip to a queue in LDS, drain the queue one pair per thread, OR the verdicts into the tile's 64 row words.
__global__ __launch_bounds__(kSupThreads) void rotated_sup_matrix_kernel(const SupArgs a) {
  __shared__ float2 s_clip[(kSupThreads / kWave) * kClipSlotsPerWave];
  constexpr uint32_t kRows = kSupRows, kSlices = 64 / kRows;   // rows of the tile this workgroup takes
  __shared__ float s_rows[kRows * 6], s_cols[64 * 6], s_rcls[kRows];
  __shared__ uint32_t s_words[kRows * 2];
  __shared__ uint16_t s_queue[kRows * 64];
  __shared__ uint32_t s_count;
  const uint32_t nblk = a.m_max / 64, nblk_launch = a.m_launch / 64;
  const uint32_t row0 = (blockIdx.x % kSlices) * kRows;      // first row of the slice inside its tile
  uint32_t t = blockIdx.x / kSlices, bi = 0;
  while (t >= nblk_launch - bi) { t -= nblk_launch - bi; ++bi; }   // upper-triangular tile index -> (row block, column block)
  const uint32_t bj = bi + t;
  const uint32_t img = blockIdx.y;
  if (bj * 64 + 64 <= a.m_done || (a.done && a.done[img])) return;   // written by the first step | image already finished
  uint32_t n = a.first_n[img];
  n = n < a.m_launch ? n : a.m_launch;
  if (bj * 64 >= n || bi * 64 + row0 >= n) return;           // (bi <= bj) nothing of this slice exists
  const int tid = static_cast<int>(threadIdx.x);
  const int lane = lane_id();
  const int wave = tid >> 6;
  const bool own_angle = (a.flags & ODTK_FLAG_ROTATED_NMS_FIXED_ANGLE) != 0;
  const float *boxes = a.first_box + static_cast<size_t>(img) * a.m_max * 6;
  const float *classes = a.first_cls + static_cast<size_t>(img) * a.m_max;
  // stage the tile's 64 row boxes and 64 column boxes (6 floats each: 384 + 384 values, coalesced)
  for (int e = tid; e < 64 * 6; e += kSupThreads) {
    const uint32_t ri = (bi * 64 + row0) * 6 + e, ci = bj * 64 * 6 + e;
    if (e < static_cast<int>(kRows) * 6) s_rows[e] = ri < n * 6 ? boxes[ri] : 0.0f;
    s_cols[e] = ci < n * 6 ? boxes[ci] : 0.0f;
  }
  if (tid < static_cast<int>(kRows)) s_rcls[tid] = bi * 64 + row0 + tid < n ? classes[bi * 64 + row0 + tid] : __builtin_nanf("");
  if (tid < static_cast<int>(kRows) * 2) s_words[tid] = 0;
  if (tid == 0) s_count = 0;
  const uint32_t j = bj * 64 + static_cast<uint32_t>(lane);
  const float jc = j < n ? classes[j] : __builtin_nanf("");   // NaN equals nothing
  __syncthreads();
  float jb[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) jb[k] = s_cols[lane * 6 + k];
  constexpr uint32_t kRowsPerWave = kRows / (kSupThreads / kWave);
  for (uint32_t r = 0; r < kRowsPerWave; ++r) {
    const uint32_t il = static_cast<uint32_t>(wave) * kRowsPerWave + r, i = bi * 64 + row0 + il;
    bool need = j > i && jc == s_rcls[il];                   // (i >= n: its class is NaN)
    if (need) need = !rotated_far_apart(s_rows + il * 6, jb, a.thresh, own_angle);
    const uint32_t slot = wave_append_slot(&s_count, need);
    if (need) s_queue[slot] = static_cast<uint16_t>((il << 6) | static_cast<uint32_t>(lane));
  }
  __syncthreads();
  const uint32_t n_q = s_count;
  float2 *clip = s_clip + static_cast<size_t>(wave) * kClipSlotsPerWave + lane;
  for (uint32_t q = static_cast<uint32_t>(tid); q < n_q; q += kSupThreads) {
    const uint32_t entry = s_queue[q], il = entry >> 6, jl = entry & 63u;
    float ib[6], cb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { ib[k] = s_rows[il * 6 + k]; cb[k] = s_cols[jl * 6 + k]; }
    if (rotated_suppresses(ib, cb, a.thresh, own_angle, clip)) atomicOr(&s_words[2 * il + (jl >> 5)], 1u << (jl & 31u));
  }
  __syncthreads();
  if (tid < static_cast<int>(kRows) && bi * 64 + row0 + tid < n)
    a.sup[(static_cast<size_t>(img) * a.m_max + bi * 64 + row0 + tid) * nblk + bj] =
        static_cast<unsigned long long>(s_words[2 * tid]) | (static_cast<unsigned long long>(s_words[2 * tid + 1]) << 32);
}

// LDS carve-up shared by host (size) and device (pointers); every offset is 16-byte aligned.  The regions of fixed size
// come first, at compile-time offsets (no scalar register per pointer -- the kernel is short of them); the ones that
// depend on detections_per_im / count follow.
template <int NB>
struct NmsFixedLds {
  static constexpr size_t sel = 0;                                             // the round's keys, in order
  static constexpr size_t box = sel + kNmsRound * 8;                           // ... their boxes
  static constexpr size_t cls = box + static_cast<size_t>(kNmsRound) * NB * 4;   // ... and classes
  static constexpr size_t hist = cls + kNmsRound * 4;                          // range_threshold's histogram | sorted-run windows | pair queue
  static constexpr size_t misc = hist + kRadixBins * 4;
  static constexpr size_t sup = misc + kNmsMisc * 4;                           // suppression words of the current chunk and the next
  static constexpr size_t end = sup + 2 * kNmsChunk * 8;
  static_assert(end % 16 == 0, "16-byte aligned regions");
};

struct NmsLds {
  static constexpr size_t kLdsBudget = 160 * 1024;
  size_t kbox, kcls, kscore, ksrc, keys, clip, total;
  int ways;     // waves that evaluate box pairs
  __host__ __device__ NmsLds(uint32_t count, int ndet, int nb, bool global_keys = false) {
    auto up = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    size_t o = nb == 6 ? NmsFixedLds<6>::end : NmsFixedLds<4>::end;
    kbox = o;   o += up(static_cast<size_t>(ndet) * nb * 4);
    kcls = o;   o += up(static_cast<size_t>(ndet) * 4);
    kscore = o; o += up(static_cast<size_t>(ndet) * 4);
    ksrc = o;   o += up(static_cast<size_t>(ndet) * 4);
    keys = o;   o += global_keys ? 0 : up(static_cast<size_t>(count) * 8);
    // rotated IoU: one lane-private polygon region (4 KiB, rotated_iou.hpp) per wave that evaluates pairs: as many of
    // the 16 waves as the 160 KiB budget allows
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

// Axis-aligned: does any box kept at ranks q0, q0 + step, ... (< q1) suppress candidate (jb, jc)?  Class words are
// fetched eight at a time (independent LDS reads) -- the common case is "no kept box of this class",
// and a one-read-per-trip loop would pay the LDS latency once per kept box.
__device__ __forceinline__ bool pull_against_kept(const float *s_kcls, const float *s_kbox, int q0, int q1, int step,
                                                  const float *jb, float jc, bool alive, float thr) {
  constexpr int kBatch = 8;
  for (int q = q0; q < q1 && alive; q += kBatch * step) {
    float kc[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) kc[u] = (q + u * step < q1) ? s_kcls[q + u * step] : __builtin_nanf("");   // NaN equals nothing
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (alive && kc[u] == jc) {                               // box.py:351: a different class keeps
        float mb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mb[k] = s_kbox[(q + u * step) * 4 + k];
        if (axis_suppresses(mb, jb, thr)) alive = false;
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

// #{i < n : A[i] > x} (kStrict) or #{A[i] >= x} for a descending-sorted A, by binary lifting with a block-uniform trip
// count (p2 = a power of two >= n): every lane runs the same log2(p2) + 1 steps whatever its n.
template <bool kStrict>
__device__ __forceinline__ uint32_t count_above(const uint64_t *A, uint32_t n, uint64_t x, uint32_t p2) {
  uint32_t lo = 0;
  for (uint32_t s = p2; s > 0; s >>= 1) {
    const uint32_t at = lo + s;
    if (at <= n) {
      const uint64_t v = A[at - 1];
      if (kStrict ? v > x : v >= x) lo = at;
    }
  }
  return lo;
}

// kGlobalKeys: more candidates than the LDS holds (count > ODTK_MAX_NMS_COUNT): the key list of an image lives in the
// caller's workspace instead; rounds then walk it out of L2 -- slower, same result.
// kStage (rotated only): 0 = the whole NMS in this launch; 1 = export the first round and stop; 2 = resolve the first round
// from the suppression matrix, then go on as stage 0 would.
template <int NB, bool kGlobalKeys = false, int kStage = 0>
__global__ __launch_bounds__(kNmsThreads) void nms_kernel(const NmsArgs a) {
  static_assert(kStage == 0 || NB == 6, "the staged form exists for rotated boxes only");
  if constexpr (kStage == 2) {
    if (a.step == 2 && a.done[blockIdx.x]) return;           // (block-uniform) step 1 finished this image: leave at once
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const NmsLds lay(a.count, a.ndet, NB, kGlobalKeys);
  uint64_t *s_keys = kGlobalKeys ? a.key_scratch + static_cast<size_t>(blockIdx.x) * a.count
                                 : reinterpret_cast<uint64_t *>(smem + lay.keys);
  using Fixed = NmsFixedLds<NB>;
  uint64_t *s_sel = reinterpret_cast<uint64_t *>(smem + Fixed::sel);
  float *s_box = reinterpret_cast<float *>(smem + Fixed::box);
  float *s_cls = reinterpret_cast<float *>(smem + Fixed::cls);
  float *s_kbox = reinterpret_cast<float *>(smem + lay.kbox);
  float *s_kcls = reinterpret_cast<float *>(smem + lay.kcls);
  float *s_kscore = reinterpret_cast<float *>(smem + lay.kscore);
  int32_t *s_ksrc = reinterpret_cast<int32_t *>(smem + lay.ksrc);
  uint32_t *s_hist = reinterpret_cast<uint32_t *>(smem + Fixed::hist);
  uint32_t *s_misc = reinterpret_cast<uint32_t *>(smem + Fixed::misc);
  // s_misc: [0..31] range_threshold scratch, [32] length of the key list, [33] gather cursor, [34] kept count,
  //         [35..36] pair-queue lengths (two, used alternately), [37] survivors of a filter pass, [38..39] boxes kept in the chunk,
  //         [40..71] verdict words of the axis-aligned pull (16 x 64 bits); [40..103] min / max keys per wave (generic mode,
  //         between rounds), [72..111] per-run valid counts, cursors, member counts, probe keys (sorted-run mode),
  //         [112..143] dead bits of the round's 1024 candidates (rotated), [144..147] smallest / largest key of the list,
  //         [148..155] output pointers
  uint64_t *s_alive = reinterpret_cast<uint64_t *>(s_misc + 40);
  uint32_t *s_dead = s_misc + 112;
  uint64_t *s_sup = reinterpret_cast<uint64_t *>(smem + Fixed::sup);
  float2 *s_clip = reinterpret_cast<float2 *>(smem + lay.clip);     // rotated only
  uint64_t *s_mat = reinterpret_cast<uint64_t *>(smem + lay.clip);  // stage 2: the suppression matrix (until the first polygon is clipped)
  uint64_t *s_win = reinterpret_cast<uint64_t *>(s_hist);           // sorted-run mode: the runs' probed slots
  uint32_t *s_queue = s_hist;                                       // rotated: pair queue (chunk phase; never while a round is selected)
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
  const float *in_s = uniform_ptr(a.scores + static_cast<size_t>(img) * count);
  const float *in_b = uniform_ptr(a.boxes + static_cast<size_t>(img) * count * NB);
  const float *in_c = uniform_ptr(a.classes + static_cast<size_t>(img) * count);
  auto stamp = [&](int k) { if (a.trace && tid == 0) a.trace[(gridDim.x + blockIdx.x) * 8 + k] = wall_clock64(); };
  // debug: image 0's phases beyond the first round (id, time) pairs: 1 round selected, 2 boxes staged, 3 chunks / push done,
  // 4 filter done, 5 push over everything done
  uint32_t n_phase = 0;
  auto phase = [&](unsigned long long id) {
    if constexpr (NB == 4)                                   // (the rotated kernels have no register to spare for it)
    if (a.trace && tid == 0 && blockIdx.x == 0 && n_phase < 48) {
      a.trace[4096 - 8 * 64 + 96 + 2 * n_phase] = id;
      a.trace[4096 - 8 * 64 + 97 + 2 * n_phase] = wall_clock64();
      ++n_phase;
    }
  };

  if (tid == 0) { s_misc[32] = 0; s_misc[34] = 0; s_misc[35] = 0; s_misc[36] = 0; }
  // the output pointers are needed once, at the very end: parked in LDS (s_misc[148..155]) they do not occupy scalar
  // registers -- which this kernel is short of -- for its whole run
  uint64_t *s_out = reinterpret_cast<uint64_t *>(s_misc + 148);
  if (tid == 0) {
    s_out[0] = reinterpret_cast<uint64_t>(a.out_scores); s_out[1] = reinterpret_cast<uint64_t>(a.out_boxes);
    s_out[2] = reinterpret_cast<uint64_t>(a.out_classes); s_out[3] = reinterpret_cast<uint64_t>(a.out_indices);
  }
  // Sorted-run mode (odtk_detect): the input is what decode_levels wrote -- n_runs lists of run_len candidates, each already
  // in NMS order, `run_valid` of them with a positive score.  The best candidates overall are then prefixes of the runs: no
  // key list, no min / max pass, no radix selection, no sort (until a filter pass turns what is left into a key list).
  static_assert(ODTK_MAX_LEVELS <= 8, "the sorted-run state (s_misc[72..111]) holds 8 runs; run_len is set by odtk_detect only");
  const uint32_t n_runs = a.run_len ? count / a.run_len : 0;
  bool runs = a.run_valid && a.run_len >= 64 && n_runs * a.run_len == count && n_runs >= 1 && n_runs <= 8;   // block-uniform
  uint32_t *s_valid = s_misc + 72, *s_cursor = s_misc + 80, *s_members = s_misc + 88;   // per run (s_misc[72..95])
  uint64_t *s_probe = reinterpret_cast<uint64_t *>(s_misc + 96);                       // per run (s_misc[96..111])
  uint32_t list_n = 0;               // generic mode: keys in s_keys; their range [k_lo, k_hi] is parked in s_misc[144..147]
  uint64_t *s_range = reinterpret_cast<uint64_t *>(s_misc + 144);
  // smallest / largest key of the list: the round selection cuts THAT range (fp32 scores of one image share their exponent
  // bits; an MSD digit needed two passes, 9.9 us, to isolate the first 256..1024 keys)
  auto key_range = [&]() {
    uint64_t k_lo = ~0ull, k_hi = 0;
    for (uint32_t i = tid; i < list_n; i += kNmsThreads) {
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
    __syncthreads();                                         // (s_misc[40..103] may still be read as verdict words)
    if (lane == 0) { s_alive[wave] = k_lo; s_alive[16 + wave] = k_hi; }
    __syncthreads();
    for (int w = 0; w < kNmsThreads / kWave; ++w) {
      k_lo = s_alive[w] < k_lo ? s_alive[w] : k_lo;
      k_hi = s_alive[16 + w] > k_hi ? s_alive[16 + w] : k_hi;
    }
    if (tid == 0) { s_range[0] = k_lo; s_range[1] = k_hi; }
    __syncthreads();
  };

  uint32_t left = 0;                 // candidates not yet handed to a round (block-uniform)
  uint32_t my_valid = 0;
  if (runs && tid < 8 && static_cast<uint32_t>(tid) < n_runs) my_valid = a.run_valid[static_cast<size_t>(img) * n_runs + tid];
  if constexpr (kStage == 2) {
    // Stage 1 selected and ordered the first round already: its keys, the boxes / classes of the candidates the matrix covers
    // (stage 1 exported them in order: coalesced, and not behind the keys as a gather is) and the matrix itself are requested
    // HERE, together with the runs' lengths: ONE global latency for everything the first round needs (round 3: lengths ->
    // state -> keys -> gather, and two more per chunk for the matrix words).  Ranks the round does not have are never read
    // (their dead bits are set); ranks >= m_max are staged when -- if -- a chunk gets there.
    const uint64_t my_key = a.first_keys[static_cast<size_t>(img) * kNmsRound + tid];
    float fb[NB], fc = 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) fb[k] = 0.0f;
    if (static_cast<uint32_t>(tid) < a.m_max) {
#pragma unroll
      for (int k = 0; k < NB; ++k) fb[k] = a.first_box[(static_cast<size_t>(img) * a.m_max + tid) * NB + k];
      fc = a.first_cls[static_cast<size_t>(img) * a.m_max + tid];
    }
    // the rows of the matrix this step can use, as they lie in memory (row stride m_max / 64 words): a flat, coalesced copy
    const uint32_t n_vec = (a.step == 1 ? a.m_first : a.m_max) * (a.m_max / 64) / 2;
    const vuint4 *sup = reinterpret_cast<const vuint4 *>(a.sup + static_cast<size_t>(img) * a.m_max * (a.m_max / 64));
    vuint4 *dst = reinterpret_cast<vuint4 *>(s_mat);
    for (uint32_t e = tid; e < n_vec; e += kNmsThreads) dst[e] = sup[e];
    s_sel[tid] = my_key;
#pragma unroll
    for (int k = 0; k < NB; ++k) s_box[tid * NB + k] = fb[k];
    s_cls[tid] = fc;
  }
  if (runs) {
    if (tid < 8) {
      const uint32_t v = my_valid;
      s_valid[tid] = v < a.run_len ? v : a.run_len;
      s_cursor[tid] = 0;
    }
    __syncthreads();
    for (uint32_t l = 0; l < n_runs; ++l) left += s_valid[l];
    left = __builtin_amdgcn_readfirstlane(left);
  } else {
    __syncthreads();
    // ---- compact positive-score candidates into 64-bit (score, ~position) keys ----
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
        if (static_cast<uint32_t>(u) * kNmsThreads < count) compact(my_scores[u], u * kNmsThreads + tid);   // block-uniform
      }
    }
    __syncthreads();
    list_n = __builtin_amdgcn_readfirstlane(s_misc[32]);   // (LDS values are VGPRs to the compiler: pin loop-control scalars to SGPRs)
    left = list_n;
  }
  const uint32_t K = left;
  stamp(1);
  if constexpr (kStage == 1) {
    if (left == 0) {                                       // no candidate: an empty first round
      if (tid == 0) a.first_n[img] = 0;
      return;
    }
  }
  if (!runs) key_range();

  uint32_t examined = 0;             // candidates handed to rounds so far (debug trace)
  uint64_t upper = ~0ull;            // generic mode: keys >= upper were consumed
  int kept = 0;                      // block-uniform copy of s_misc[34]
  int pulled = 0;                    // every candidate still listed is known to survive the boxes kept at ranks < pulled
  bool first_round = true;
  bool push_all_done = false;        // the push over everything has had its kMaxPushAll boxes (round 6)
  int last_round_kept = 0;           // yield of the previous round (push / pull choice)
  uint32_t last_round_size = 0;

  // ---- rotated: box pairs through the queue (the ONE place of the kernel that clips polygons) ----
  // Candidates are the round's ranks r0 .. r0 + n - 1 (boxes / classes staged in s_box / s_cls); `rows` = false: against
  // the kept boxes at ranks ka .. kb - 1, a suppressed candidate gets its dead bit; `rows` = true (n <= 64): candidate
  // pairs (i < j) inside the chunk, both alive, the verdict is bit j of suppression row i.  Slabs of kPairQueue (box, box)
  // combinations are filtered (class, dead bit, distance reject) into the queue, then drained one pair per thread.
  uint32_t q_parity = 0;
  auto pair_phase = [&](bool rows, uint32_t r0, uint32_t n, int ka, int kb) {
    if constexpr (NB == 6) {
      const uint32_t n_a = rows ? n : static_cast<uint32_t>(kb - ka);
      const uint32_t combos = n_a * n;
      float2 *clip = s_clip + static_cast<size_t>(wave) * kClipSlotsPerWave + lane;
      for (uint32_t e0 = 0; e0 < combos; e0 += kPairQueue, q_parity ^= 1u) {
        const uint32_t parity = q_parity;
        // two queue counters, used by consecutive slabs in turn -- across calls too: a slab clears the one its successor will
        // use (last read before the previous slab's closing barrier, next touched after this slab's barriers)
        if (tid == 0) s_misc[35 + (parity ^ 1u)] = 0;
#pragma unroll
        for (int u = 0; u < kPairQueue / kNmsThreads; ++u) {
          const uint32_t e = e0 + static_cast<uint32_t>(u) * kNmsThreads + tid;
          const uint32_t ai = e / n, ji = e - ai * n;          // (n = 64 or 1024 on the hot paths: `ai` is wave-uniform there)
          const uint32_t r = r0 + ji;
          bool want = e < combos && !((s_dead[r >> 5] >> (r & 31u)) & 1u);
          uint32_t entry = 0;
          if (want) {
            const float *jb = s_box + static_cast<size_t>(r) * 6;
            if (rows) {
              const uint32_t ri = r0 + ai;
              want = ai < ji && !((s_dead[ri >> 5] >> (ri & 31u)) & 1u) && s_cls[ri] == s_cls[r] &&
                     !rotated_far_apart(s_box + static_cast<size_t>(ri) * 6, jb, thr, own_angle);
              entry = 0x80000000u | (ri << 10) | r;
            } else {
              const uint32_t k = static_cast<uint32_t>(ka) + ai;
              want = s_kcls[k] == s_cls[r] && !rotated_far_apart(s_kbox + static_cast<size_t>(k) * 6, jb, thr, own_angle);
              entry = (k << 10) | r;
            }
          }
          const uint32_t slot = wave_append_slot(&s_misc[35 + parity], want);
          if (want) s_queue[slot] = entry;
        }
        __syncthreads();
        const uint32_t n_q = s_misc[35 + parity];
        if (wave < ways) {
          for (uint32_t q = static_cast<uint32_t>(tid); q < n_q; q += static_cast<uint32_t>(ways) * kWave) {
            const uint32_t entry = s_queue[q];
            const uint32_t r = entry & 1023u, ai = (entry >> 10) & 2047u;
            const bool is_row = (entry >> 31) != 0;
            const float *mp = (is_row ? s_box : s_kbox) + static_cast<size_t>(ai) * 6;
            float mb[6], jb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { mb[k] = mp[k]; jb[k] = s_box[static_cast<size_t>(r) * 6 + k]; }
            if (rotated_suppresses(mb, jb, thr, own_angle, clip)) {
              if (is_row) {
                const uint32_t bit = r - r0;                 // row ai - r0 (< 64), column r - r0
                atomicOr(reinterpret_cast<uint32_t *>(s_sup + (ai - r0)) + (bit >> 5), 1u << (bit & 31u));
              } else {
                atomicOr(&s_dead[r >> 5], 1u << (r & 31u));
              }
            }
          }
        }
        __syncthreads();
      }
    }
  };

  while (left > 0 && kept < ndet) {
    // ---- heavy suppression, everything that is left fits two candidates per thread (axis-aligned): push over ALL of it ----
    // After a filter pass the survivors are a key list, and when the last round kept fewer than 1 in 20 of its candidates
    // most of them will die too.  No round, no sort, no further filter then: every thread owns up to two candidates (key,
    // box, class in registers), and per KEPT box there is one block-wide maximum over the alive keys (the greedy order IS
    // "best alive candidate next") and one parallel test of every alive candidate against it -- ~0.7 us per kept box, where a
    // round costs ~10 us of selection and sorting before its first box.
    if constexpr (NB == 4) {
      // (round 6: the push costs ~2 us per box it KEEPS -- one block-wide maximum, two barriers and an IoU block per owned slot in
      //  every wave -- so it pays only while few more boxes are to come.  How many will come cannot be read off the first round's
      //  yield (its candidates are the duplicates of the best objects; a trained detector's crowded image kept 4 of its first 250
      //  and 80 of the 1500 behind them: 168 us here), so the push is given kMaxPushAll boxes: the pathological RN101 heads -- 9
      //  kept of 2236, five of them here, 14 us -- finish inside it, anything that still has candidates alive after them leaves
      //  with its survivors compacted and goes through sorted rounds of batched pushes.  profiles/r06_nms_clustered.txt)
      constexpr int kMaxPushAll = 8;
      // (kOwn = 2 since the bail-out exists: four candidates per thread -- 28 registers -- pushed the 1024-thread kernel over its 128
      //  and into scratch; the case this form is for, the RN101 heads' ~1100 survivors of a filter pass, fits two per thread)
      constexpr int kOwn = 2;
      if (!runs && !first_round && !push_all_done && left <= static_cast<uint32_t>(kOwn) * kNmsThreads && list_n <= static_cast<uint32_t>(kOwn) * kNmsThreads &&
          static_cast<uint32_t>(last_round_kept) * 20u < last_round_size) {
        uint64_t mk[kOwn];
        float mb[kOwn][4], mc[kOwn];
#pragma unroll
        for (int u = 0; u < kOwn; ++u) {
          const uint32_t i = static_cast<uint32_t>(u) * kNmsThreads + tid;
          uint64_t key = i < list_n ? s_keys[i] : 0;
          key = key < upper ? key : 0;                         // already handed to a round
          mk[u] = key;
          mc[u] = 0.0f;
#pragma unroll
          for (int k = 0; k < 4; ++k) mb[u][k] = 0.0f;
          if (key) {
            const uint32_t p = key_index(key);
#pragma unroll
            for (int k = 0; k < 4; ++k) mb[u][k] = in_b[static_cast<size_t>(p) * 4 + k];
            mc[u] = in_c[p];
          }
        }
        if (kept > pulled) {                                   // (the boxes kept since the last filter pass)
#pragma unroll
          for (int u = 0; u < kOwn; ++u)
            if (mk[u] && !pull_against_kept(s_kcls, s_kbox, pulled, kept, 1, mb[u], mc[u], true, thr)) mk[u] = 0;
        }
        int pushes = 0;
        bool bailed = false;
        while (kept < ndet) {                                  // block-uniform trip count
          uint64_t best = wave_max_u64(mk[0] > mk[1] ? mk[0] : mk[1]);   // (DPP row shifts / broadcasts: no LDS crossbar round trips)
          static_assert(kOwn == 2, "the maximum above is written for two");
          if (lane == 0) s_alive[wave] = best;
          __syncthreads();
          best = 0;
          for (int w = 0; w < kNmsThreads / kWave; ++w) best = s_alive[w] > best ? s_alive[w] : best;
          best = uniform_u64(best);
          if (best == 0) { __syncthreads(); break; }           // nothing alive
#pragma unroll
          for (int u = 0; u < kOwn; ++u) {
            if (mk[u] == best) {                               // its owner keeps it (keys are unique) and shows its box to everybody
#pragma unroll
              for (int k = 0; k < 4; ++k) { s_kbox[kept * 4 + k] = mb[u][k]; s_box[k] = mb[u][k]; }
              s_kcls[kept] = mc[u];
              s_cls[0] = mc[u];
              s_kscore[kept] = key_score(best);
              s_ksrc[kept] = static_cast<int32_t>(key_index(best));
              mk[u] = 0;
            }
          }
          __syncthreads();
          float kb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) kb[k] = s_box[k];
          const float kc = s_cls[0];
#pragma unroll
          for (int u = 0; u < kOwn; ++u)
            if (mk[u] && mc[u] == kc && axis_suppresses(kb, mb[u], thr)) mk[u] = 0;
          ++kept;
          if (!(a.flags & kNmsFlagChunks) && ++pushes == kMaxPushAll && kept < ndet) { bailed = true; break; }   // (block-uniform)
        }
        if (tid == 0) { s_misc[34] = static_cast<uint32_t>(kept); s_misc[37] = 0; }
        if (bailed) {
          // the candidates still alive (each has met every kept box) go back into the key list, compacted; the rounds below take over
          __syncthreads();                                     // (every owner has read its slots of s_keys; s_misc[37] is cleared)
#pragma unroll
          for (int u = 0; u < kOwn; ++u) {
            const bool live = mk[u] != 0;
            const uint32_t slot = wave_append_slot(&s_misc[37], live);
            if (live) s_keys[slot] = mk[u];
          }
          __syncthreads();
          examined += left;
          list_n = __builtin_amdgcn_readfirstlane(s_misc[37]);
          examined -= list_n;
          left = list_n;
          upper = ~0ull;
          pulled = kept;
          push_all_done = true;
          if (left > 0) key_range();
          phase(5);
          continue;
        }
        examined += left;
        left = 0;
        phase(5);
        break;
      }
    }
    const int kept_at_round_start = kept;
    // ---- round: the next (up to) 1024 best candidates, in order, into s_sel ----
    uint32_t n_round = left < kNmsRound ? left : kNmsRound;
    bool staged = false;               // the round's boxes / classes are in s_box / s_cls already (block-uniform)
    if (kStage == 2 && first_round) {
      // stage 1 selected and ordered this round already (its keys, boxes and matrix are in LDS since the kernel's first
      // instructions): its size and the state of the selection after it
      const uint32_t *state = a.first_state + static_cast<size_t>(img) * 16;
      n_round = __builtin_amdgcn_readfirstlane(state[0]);
      if (runs) { if (tid < 8) s_cursor[tid] = state[1 + tid]; }
      else upper = uniform_u64(static_cast<uint64_t>(state[1]) | (static_cast<uint64_t>(state[2]) << 32));
      __syncthreads();
    } else if (runs && kStage == 1) {
      // Rotated, stage 1 (= the first round: every cursor is zero).  The round should be as long as the suppression matrix
      // (m_max, several hundred candidates), and the probe of 1024 / n_runs slots per run (below) returns as few as
      // 1024 / n_runs when one level dominates.  Exact top-`want` (want = min(left, m_max)) instead, in two steps on the
      // runs' first min(valid, 1024) keys, loaded ONCE, window after window, into the key list's space:
      //   cut    wave q <-> run q: 64 evenly spaced sample keys of the window, each ranked against all windows (one binary
      //          search per run, advanced together); ranks are monotone along a run, so the first sample with rank >= want
      //          bounds what the run can contribute: its prefix [0, u_q); the prefixes hold want + n_runs x 16 keys at most
      //   rank   thread <-> key of a prefix: its exact rank (the same searches, over the prefixes only); rank < want -> it is
      //          in the round, at position `rank`.
      // Round 3 ranked ALL window keys (~5000) against all windows: 21 us; a first attempt this round repeated the 1024 / n_runs
      // probe on the windows until the round was long enough: 42 us when one level dominates (8 probes).
      uint32_t *s_wbase = s_misc + 96, *s_upto = s_misc + 104;   // (the probe keys of the generic form are not used here)
      if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t q = 0; q < 8; ++q) {
          s_wbase[q] = acc;
          const uint32_t v = q < n_runs ? s_valid[q] : 0u;
          acc += v < kNmsRound ? v : kNmsRound;
        }
        s_misc[33] = acc;
      }
      __syncthreads();
      const uint32_t total = __builtin_amdgcn_readfirstlane(s_misc[33]);
      for (uint32_t f = tid; f < total; f += kNmsThreads) {
        uint32_t q = 0;
#pragma unroll
        for (uint32_t t = 1; t < 8; ++t) q += (t < n_runs && f >= s_wbase[t]) ? 1u : 0u;
        const uint32_t p = q * a.run_len + (f - s_wbase[q]);
        s_keys[f] = make_key(in_s[p], p);
      }
      __syncthreads();
      const uint32_t want = left < a.m_max ? left : a.m_max;   // (<= the windows' total: a window is the whole run or 1024 keys)
      uint32_t w_n[8], w_b[8];                                 // the windows (block-uniform)
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) {
        const uint32_t v = q < n_runs ? s_valid[q] : 0u;
        w_n[q] = v < kNmsRound ? v : kNmsRound;
        w_b[q] = w_n[q] ? s_wbase[q] : 0u;
      }
      // #{keys of the windows' first lim[.] entries above `key`}: the binary searches of all runs advance together
      auto rank_of = [&](uint64_t key, const uint32_t *lim, uint32_t p2) {
        uint32_t lo[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) lo[q] = 0;
        for (uint32_t s2 = p2; s2 > 0; s2 >>= 1) {
#pragma unroll
          for (uint32_t q = 0; q < 8; ++q) {
            if (q < n_runs) {                                  // (block-uniform)
              const uint32_t at = lo[q] + s2;
              const bool in_range = at <= lim[q];
              const uint64_t v = s_keys[w_b[q] + (in_range ? at - 1 : 0u)];
              lo[q] = in_range && v > key ? at : lo[q];
            }
          }
        }
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) rank += lo[q];
        return rank;
      };
      if (static_cast<uint32_t>(wave) < n_runs) {              // cut
        uint32_t my_n = 0, my_b = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) { my_n = q == static_cast<uint32_t>(wave) ? w_n[q] : my_n; my_b = q == static_cast<uint32_t>(wave) ? w_b[q] : my_b; }
        uint32_t upto = 0;
        if (my_n) {                                            // (wave-uniform)
          const uint32_t stride = (my_n + 63u) / 64u;
          uint32_t pos = (static_cast<uint32_t>(lane) + 1u) * stride;
          pos = (pos < my_n ? pos : my_n) - 1u;                // the window's last key closes the samples
          const bool in = rank_of(s_keys[my_b + pos], w_n, kNmsRound) < want;
          const uint32_t n_in = static_cast<uint32_t>(__popcll(__ballot(in)));   // monotone: the first n_in samples
          // the first sample that is out sits at (n_in + 1) * stride - 1 (or is the window's last key): nothing from there on
          const uint32_t first_out = (n_in + 1u) * stride - 1u;
          upto = n_in == 64u ? my_n : (first_out < my_n ? first_out : my_n - 1u);
        }
        if (lane == 0) s_upto[wave] = upto;
      }
      __syncthreads();
      uint32_t u_n[8], u_base[9];
      u_base[0] = 0;
      uint32_t longest = 0;
#pragma unroll
      for (uint32_t q = 0; q < 8; ++q) {
        u_n[q] = q < n_runs ? __builtin_amdgcn_readfirstlane(s_upto[q]) : 0u;
        u_base[q + 1] = u_base[q] + u_n[q];
        longest = u_n[q] > longest ? u_n[q] : longest;
      }
      uint32_t p2 = 1;
      while (p2 < longest) p2 <<= 1;
      if (static_cast<uint32_t>(tid) < u_base[8]) {            // rank (u_base[8] <= want + 16 n_runs <= 1024)
        uint32_t q = 0, base = 0;
#pragma unroll
        for (uint32_t t = 1; t < 8; ++t) {
          const bool past = static_cast<uint32_t>(tid) >= u_base[t];
          q = past ? t : q;
          base = past ? u_base[t] : base;
        }
        uint32_t my_b = 0;
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) my_b = t == q ? w_b[t] : my_b;
        const uint64_t key = s_keys[my_b + (static_cast<uint32_t>(tid) - base)];
        // its box and class are requested before the rank is known (cursors are zero: the key sits at its offset in run q),
        // so the gather overlaps the searches instead of following them
        const uint32_t pos = q * a.run_len + (static_cast<uint32_t>(tid) - base);
        float pb[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) pb[k] = in_b[static_cast<size_t>(pos) * NB + k];
        const float pc = in_c[pos];
        const uint32_t rank = rank_of(key, u_n, p2);
        if (rank < want) {
          s_sel[rank] = key;
#pragma unroll
          for (int k = 0; k < NB; ++k) s_box[rank * NB + k] = pb[k];
          s_cls[rank] = pc;
        }
      }
      staged = true;
      __syncthreads();
      if (static_cast<uint32_t>(tid) < n_runs) {               // how far every run was consumed: its keys >= the round's last key
        uint32_t n_w = 0, b_w = 0;
#pragma unroll
        for (uint32_t t = 0; t < 8; ++t) { n_w = t == static_cast<uint32_t>(tid) ? u_n[t] : n_w; b_w = t == static_cast<uint32_t>(tid) ? w_b[t] : b_w; }
        s_cursor[tid] += count_above<false>(s_keys + b_w, n_w, s_sel[want - 1], p2);
      }
      __syncthreads();
      n_round = want;
    } else if (runs) {
      // slot t = (run l, offset j): the next `step` candidates of every run.  T = the largest of the runs' LAST examined
      // keys; the members of the round are the slots with key >= T: no run can hold an unexamined key >= T (its last
      // examined key is <= T and the run is sorted), so they are exactly the best unconsumed candidates, 1 .. n_runs * step of
      // them.  Keys are unique: a member's position in the round is the number of member keys above it = its own offset
      // (the run is sorted) + per other run the count of examined keys above it (all of those are >= T, i.e. members).
      uint32_t step = kNmsRound / n_runs;
      // (opaque to the optimiser: everything derived from `step` -- each thread's run, offset and addresses -- is then
      // recomputed per round instead of being hoisted out of the round loop and kept alive, in VGPRs the polygon clip
      // needs, for the whole kernel)
      asm volatile("" : "+s"(step));
      uint32_t p2 = 1;
      while (p2 < step) p2 <<= 1;
      const uint32_t l = static_cast<uint32_t>(tid) / step, j = static_cast<uint32_t>(tid) - l * step;
      if (tid < 8) s_probe[tid] = 0;
      __syncthreads();
      uint64_t key = 0;
      float pb[NB], pc = 0.0f;                               // (axis-aligned) the slot's box and class, requested WITH its score:
#pragma unroll
      for (int k = 0; k < NB; ++k) pb[k] = 0.0f;             // the position is known before the rank is -- one global latency per
      if (l < n_runs) {                                      // round instead of two (score, then a gather behind the ordered keys)
        const uint32_t avail = s_valid[l] - s_cursor[l];
        const uint32_t n_l = avail < step ? avail : step;
        if (j < n_l) {
          const uint32_t p = l * a.run_len + s_cursor[l] + j;
          key = make_key(in_s[p], p);
          if constexpr (NB == 4) {
#pragma unroll
            for (int k = 0; k < NB; ++k) pb[k] = in_b[static_cast<size_t>(p) * NB + k];
            pc = in_c[p];
          }
          if (j == n_l - 1) s_probe[l] = key;
        }
      }
      s_win[tid] = key;                                      // run l's window = s_win[l * step .. + step)
      __syncthreads();
      if (first_round) stamp(2);
      uint64_t T = 0;
      for (uint32_t q = 0; q < n_runs; ++q) T = s_probe[q] > T ? s_probe[q] : T;
      if (key != 0 && key >= T) {
        // the binary searches of all runs advance together, one step of each per trip: independent LDS reads in flight
        // instead of n_runs x log2(step) dependent ones.  The key's own run is searched like the others (the count there is
        // its own offset j): no per-thread exception, so every load of a trip is unconditional for the runs that exist.
        uint32_t n_q[8], lo[8];
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) {
          const uint32_t avail = q < n_runs ? s_valid[q] - s_cursor[q] : 0u;
          n_q[q] = avail < step ? avail : step;
          lo[q] = 0;
        }
        for (uint32_t s2 = p2; s2 > 0; s2 >>= 1) {
#pragma unroll
          for (uint32_t q = 0; q < 8; ++q) {
            if (q < n_runs) {                                  // (block-uniform)
              const uint32_t at = lo[q] + s2;
              const bool in_range = at <= n_q[q];
              const uint64_t v = s_win[q * step + (in_range ? at - 1 : 0u)];
              lo[q] = in_range && v > key ? at : lo[q];
            }
          }
        }
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; ++q) rank += lo[q];
        s_sel[rank] = key;
        if constexpr (NB == 4) {
#pragma unroll
          for (int k = 0; k < NB; ++k) s_box[rank * NB + k] = pb[k];
          s_cls[rank] = pc;
        }
      }
      staged = NB == 4;
      if (static_cast<uint32_t>(tid) < n_runs) {
        const uint32_t avail = s_valid[tid] - s_cursor[tid];
        s_members[tid] = count_above<false>(s_win + static_cast<uint32_t>(tid) * step, avail < step ? avail : step, T, p2);
      }
      __syncthreads();
      n_round = 0;
      for (uint32_t q = 0; q < n_runs; ++q) n_round += s_members[q];
      __syncthreads();
      if (static_cast<uint32_t>(tid) < n_runs) s_cursor[tid] += s_members[tid];
      n_round = __builtin_amdgcn_readfirstlane(n_round);
      if (first_round) stamp(3);
    } else {
      const LdsKeySource src{s_keys, list_n, upper};
      uint64_t lower = 0;
      // any top-prefix of 256..1024 keys will do for a round: stop the radix descent early
      if (left > kNmsRound) {
        const uint64_t k_lo = uniform_u64(s_range[0]), k_hi = uniform_u64(s_range[1]);
        lower = uniform_u64(range_threshold(src, 256, kNmsRound, k_lo, upper == ~0ull ? k_hi : upper - 1, s_hist, s_misc, &n_round));
      }
      if (tid == 0) s_misc[33] = 0;
      __syncthreads();
      for (uint32_t i0 = 0; i0 < list_n; i0 += kNmsThreads) {     // (block-uniform trip count: the append ballots)
        const uint32_t i = i0 + tid;
        const uint64_t key = i < list_n ? s_keys[i] : 0;
        const bool take = key != 0 && key < upper && key >= lower;
        const uint32_t slot = wave_append_slot(&s_misc[33], take);
        if (take) s_sel[slot] = key;
      }
      __syncthreads();
      if (first_round) stamp(2);
      n_round = __builtin_amdgcn_readfirstlane(n_round);
      sort_keys_desc(s_sel, n_round);                        // pads to 1024 with zeros (sort last)
      if (first_round) stamp(3);
      upper = lower;
    }
    left -= n_round;
    examined += n_round;
    phase(1);

    // thread t <-> rank `t` of this round: stage its box + class in LDS
    if (static_cast<uint32_t>(tid) < n_round && !(kStage == 2 && first_round) && !staged) {
      const uint32_t p = key_index(s_sel[tid]);
#pragma unroll
      for (int k = 0; k < NB; ++k) s_box[tid * NB + k] = in_b[static_cast<size_t>(p) * NB + k];
      s_cls[tid] = in_c[p];
    }
    // candidates of the first round the suppression matrix covers: all of them, or the first m_max (whole chunks)
    // (two-step speculation, step 1: only the first m_first of them have their rows yet)
    const uint32_t m_all = kStage != 0 && first_round ? (n_round <= a.m_max ? n_round : a.m_max) : 0u;
    const bool first_step = kStage == 2 && a.step == 1 && m_all > a.m_first;
    const uint32_t m_cov = first_step ? a.m_first : m_all;
    if constexpr (kStage == 1) {
      if (static_cast<uint32_t>(tid) < m_cov) {
#pragma unroll
        for (int k = 0; k < NB; ++k) a.first_box[(static_cast<size_t>(img) * a.m_max + tid) * NB + k] = s_box[tid * NB + k];
        a.first_cls[static_cast<size_t>(img) * a.m_max + tid] = s_cls[tid];
      }
      if (static_cast<uint32_t>(tid) < n_round) a.first_keys[static_cast<size_t>(img) * kNmsRound + tid] = s_sel[tid];
      uint32_t *state = a.first_state + static_cast<size_t>(img) * 16;
      if (tid == 0) {
        a.first_n[img] = m_cov;
        state[0] = n_round;
        if (!runs) { state[1] = static_cast<uint32_t>(upper); state[2] = static_cast<uint32_t>(upper >> 32); }
      }
      if (runs && tid < 8) state[1 + tid] = s_cursor[tid];
      return;
    }
    if constexpr (NB == 6) {
      if (tid < 32) {                                          // dead bits: the slots behind the round's last candidate
        const uint32_t lo = static_cast<uint32_t>(tid) * 32u;
        s_dead[tid] = lo >= n_round ? ~0u : (lo + 32u <= n_round ? 0u : ~0u << (n_round - lo));
      }
    }
    __syncthreads();

    phase(2);
    // ---- heavy suppression (axis-aligned): push instead of pull ----
    // When the last round kept fewer than 1 in 20 of its candidates, resolving 64 candidates per chunk (pairwise rows +
    // two barriers, ~2.7 us) spends most of its time on candidates that die anyway.  Then: thread <-> candidate for the WHOLE
    // round, and per KEPT box one parallel pass -- the first candidate still alive is kept and every alive candidate behind
    // it of its class tests itself against it (~0.6 us per kept box: one LDS-broadcast box, one IoU, one ballot).
    bool pushed = false;
    // ---- axis-aligned, round 6: BATCHED PUSH -- the round is resolved 64 ALIVE candidates at a time ----
    // What a trained detector hands the NMS is not what the synthetic heads of the bench are: clusters of dozens of overlapping
    // same-class candidates around every object, 50-100 % of all candidates examined, 2-100 kept (profiles/r06_trained_ap.txt:
    // 20-240 us per image with the chunk loop below, which walks the round 64 RANKS at a time, dead candidates included, and
    // pays two barriers, a pull and a serial resolve per chunk for a handful of kept boxes).  Here thread <-> candidate for the
    // whole round; per trip (a) the first 64 candidates still alive are compacted (ballots + a prefix over the sixteen words),
    // (b) the sixteen waves compute their pairwise suppression rows, (c) wave 0 resolves them in rank order with scalar bit
    // arithmetic exactly as a chunk is resolved, and (d) every candidate behind the batch tests itself against the boxes the
    // batch kept -- so a kept box clears its whole cluster out of the round at once, and the next batch starts at the next
    // candidate that is really still alive.  Same verdicts in the same order as the greedy loop (a batch member has survived
    // every box kept before the batch; inside the batch the rows are resolved in rank order; everything behind is tested against
    // what the batch kept before it is looked at).  kNmsFlagChunks (ODTK_NMS_CHUNKS=1): rounds 3-5's chunk loop, A/B.
    if constexpr (NB == 4) {
      if (!(a.flags & kNmsFlagChunks) && !first_round) {     // (the first round stays with the chunk loop: on a detector that keeps most of what it examines -- one round, 100 kept of ~230 -- that form is the faster one)
        pushed = true;
        const uint32_t r = static_cast<uint32_t>(tid);
        float jb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) jb[k] = r < n_round ? s_box[r * 4 + k] : 0.0f;
        const float jc = r < n_round ? s_cls[r] : 0.0f;
        bool alive = r < n_round;
        if (alive && kept > pulled) alive = pull_against_kept(s_kcls, s_kbox, pulled, kept, 1, jb, jc, true, thr);
        uint32_t *s_bidx = reinterpret_cast<uint32_t *>(s_sup + kNmsChunk);   // the batch's ranks (the second rows buffer is idle in this mode)
        while (kept < ndet) {                                  // block-uniform trip count
          // (a) the first 64 candidates still alive, in rank order
          const uint64_t word = __ballot(alive);
          if (lane == 0) s_alive[wave] = word;
          __syncthreads();
          uint32_t before = 0, total = 0;
          for (int w = 0; w < kNmsThreads / kWave; ++w) {
            const uint32_t cw = static_cast<uint32_t>(__popcll(s_alive[w]));
            before += w < wave ? cw : 0u;
            total += cw;
          }
          total = __builtin_amdgcn_readfirstlane(total);
          if (total == 0) break;                               // (block-uniform) nobody left in this round
          const uint32_t n_b = total < static_cast<uint32_t>(kNmsChunk) ? total : static_cast<uint32_t>(kNmsChunk);
          const uint32_t slot = before + static_cast<uint32_t>(__popcll(word & ((1ull << lane) - 1ull)));
          const bool member = alive && slot < n_b;
          if (member) s_bidx[slot] = r;
          __syncthreads();
          phase(10);
          // (b) suppression rows of the batch: lane <-> batch slot j, wave w takes the rows i = w, w + 16, ...
          {
            const uint32_t rj = static_cast<uint32_t>(lane) < n_b ? s_bidx[lane] : 0u;
            float qb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) qb[k] = s_box[rj * 4 + k];
            const float qc = s_cls[rj];
            for (uint32_t i = static_cast<uint32_t>(wave); i < n_b; i += static_cast<uint32_t>(kNmsThreads / kWave)) {
              const uint32_t ri = s_bidx[i];                   // same address in every lane: LDS broadcast
              const float ic = s_cls[ri];
              const bool rival = static_cast<uint32_t>(lane) > i && static_cast<uint32_t>(lane) < n_b && qc == ic;
              uint64_t row = 0;
              if (__ballot(rival)) {                           // wave-uniform
                float ib[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) ib[k] = s_box[ri * 4 + k];
                row = __ballot(rival && axis_suppresses(ib, qb, thr));
              }
              if (lane == 0) s_sup[i] = row;
            }
          }
          __syncthreads();
          phase(11);
          // (c) wave 0 resolves the batch in rank order (scalar bit operations; the chunk loop's resolve)
          const int kept_before = kept;
          if (wave == 0) {
            const uint64_t my_row = static_cast<uint32_t>(lane) < n_b ? s_sup[lane] : 0;
            const uint32_t row_lo = static_cast<uint32_t>(my_row), row_hi = static_cast<uint32_t>(my_row >> 32);
            uint64_t mask = n_b >= 64u ? ~0ull : ((1ull << n_b) - 1ull);
            uint64_t kept_mask = 0;
            int k_cnt = kept_before;
            while (mask && k_cnt < ndet) {
              const uint64_t hot = __ballot((my_row & mask) != 0) & mask;
              uint64_t run = hot ? mask & ((hot & (0ull - hot)) - 1ull) : mask;
              const int room = ndet - k_cnt;
              while (__popcll(run) > room) run &= ~(1ull << (63 - __clzll(static_cast<long long>(run))));
              kept_mask |= run;
              k_cnt += __popcll(run);
              mask &= ~run;
              if (!hot || k_cnt >= ndet) break;
              const int l0 = __ffsll(static_cast<unsigned long long>(hot)) - 1;
              const uint64_t row = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(row_hi), l0))) << 32) |
                                   static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(row_lo), l0));
              kept_mask |= 1ull << l0;
              mask &= ~(row | (1ull << l0));
              ++k_cnt;
            }
            if ((kept_mask >> lane) & 1ull) {
              const int my_rank = kept_before + __popcll(kept_mask & ((1ull << lane) - 1ull));
              const uint32_t rr = s_bidx[lane];
              const uint64_t key = s_sel[rr];
#pragma unroll
              for (int k = 0; k < 4; ++k) s_kbox[my_rank * 4 + k] = s_box[rr * 4 + k];
              s_kcls[my_rank] = s_cls[rr];
              s_kscore[my_rank] = key_score(key);
              s_ksrc[my_rank] = static_cast<int32_t>(key_index(key));
            }
            if (lane == 0) s_misc[34] = static_cast<uint32_t>(k_cnt);
          }
          __syncthreads();
          kept = __builtin_amdgcn_readfirstlane(static_cast<int>(s_misc[34]));
          phase(12);
          // (d) the batch is settled; everybody behind it meets the boxes it kept
          if (member) alive = false;
          else if (alive && kept > kept_before) alive = pull_against_kept(s_kcls, s_kbox, kept_before, kept, 1, jb, jc, true, thr);
          phase(13);
        }
        __syncthreads();                                       // (s_alive / the batch buffers are reused by whatever follows)
      } else
      if (!first_round && static_cast<uint32_t>(last_round_kept) * 20u < last_round_size) {
        pushed = true;
        const uint32_t r = static_cast<uint32_t>(tid);
        float jb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) jb[k] = r < n_round ? s_box[r * 4 + k] : 0.0f;
        const float jc = r < n_round ? s_cls[r] : 0.0f;
        bool alive = r < n_round;
        if (alive && kept > pulled) alive = pull_against_kept(s_kcls, s_kbox, pulled, kept, 1, jb, jc, true, thr);
        while (kept < ndet) {                                  // block-uniform trip count
          const uint64_t word = __ballot(alive);
          if (lane == 0) s_alive[wave] = word;
          __syncthreads();
          uint32_t first = kNmsRound;                          // lowest alive rank of the round
          for (int w = kNmsThreads / kWave - 1; w >= 0; --w) {
            const uint64_t m = s_alive[w];
            if (m) first = static_cast<uint32_t>(w) * kWave + static_cast<uint32_t>(__ffsll(static_cast<unsigned long long>(m)) - 1);
          }
          first = __builtin_amdgcn_readfirstlane(first);
          if (first >= n_round) { __syncthreads(); break; }
          if (r == first) {                                    // keep it
            const uint64_t key = s_sel[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) s_kbox[kept * 4 + k] = jb[k];
            s_kcls[kept] = jc;
            s_kscore[kept] = key_score(key);
            s_ksrc[kept] = static_cast<int32_t>(key_index(key));
            alive = false;
          } else if (alive && r > first && jc == s_cls[first]) {
            float mb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) mb[k] = s_box[first * 4 + k];
            if (axis_suppresses(mb, jb, thr)) alive = false;
          }
          ++kept;
          __syncthreads();                                     // (s_alive is rewritten next trip)
        }
        if (tid == 0) s_misc[34] = static_cast<uint32_t>(kept);
        __syncthreads();
      }
    }

    // axis-aligned: suppression rows of the chunk at rank c_at into `rows` (row i = the lanes j > i of the same class that
    // candidate i WOULD suppress if it is kept), by waves w0, w0 + 1, ... (n_w of them; lane <-> candidate j)
    auto chunk_rows = [&](uint32_t c_at, uint64_t *rows, int w0, int n_w) {
      if constexpr (NB == 4) {
        const uint32_t rj = c_at + lane;
        const uint32_t n_c = n_round - c_at < kNmsChunk ? n_round - c_at : kNmsChunk;
        float jb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) jb[k] = s_box[rj * 4 + k];
        const float jc = s_cls[rj];
        for (uint32_t i = static_cast<uint32_t>(wave - w0); i < n_c; i += static_cast<uint32_t>(n_w)) {
          const float ic = s_cls[c_at + i];                    // same address in every lane: LDS broadcast
          const bool rival = static_cast<uint32_t>(lane) > i && rj < n_round && jc == ic;
          uint64_t row = 0;
          if (__ballot(rival)) {                               // wave-uniform: most rows have no same-class follower
            float ib[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ib[k] = s_box[(c_at + i) * 4 + k];
            row = __ballot(rival && axis_suppresses(ib, jb, thr));
          }
          if (lane == 0) rows[i] = row;
        }
      }
    };
    if constexpr (NB == 4) {
      if (!pushed && kept < ndet) chunk_rows(0, s_sup, 0, 16);   // rows of the first chunk: all waves (the pull's barrier publishes them)
    }

    // ---- chunks of 64 candidates ----
    for (uint32_t c0 = 0; c0 < n_round && kept < ndet && !pushed && !(first_step && c0 >= m_cov); c0 += kNmsChunk) {
      const int kept_before = kept;
      // debug: per-chunk timeline of image 0's first round (3 stamps per chunk: start, after the pair work, after the resolve)
      auto cstamp = [&](int k) {
        if (a.trace && tid == 0 && blockIdx.x == 0 && first_round && c0 / kNmsChunk < 20)
          a.trace[4096 - 8 * 64 + (c0 / kNmsChunk) * 4 + k] = k == 3 ? static_cast<unsigned long long>(kept) : wall_clock64();
      };
      cstamp(0);
      const uint32_t r = c0 + lane;                           // rank inside the round (< 1024)
      const uint32_t n_chunk = n_round - c0 < kNmsChunk ? n_round - c0 : kNmsChunk;
      if constexpr (kStage == 2) {
        if (first_round && c0 == a.m_max) {                    // beyond what stage 1 exported: gather the rest of the round
          const uint32_t t = static_cast<uint32_t>(tid);
          if (t >= c0 && t < n_round) {
            const uint32_t p = key_index(s_sel[t]);
#pragma unroll
            for (int k = 0; k < NB; ++k) s_box[t * NB + k] = in_b[static_cast<size_t>(p) * NB + k];
            s_cls[t] = in_c[p];
          }
          __syncthreads();
        }
      }
      if constexpr (NB == 4) {
        float jb[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) jb[k] = s_box[r * NB + k];
        const float jc = s_cls[r];
        // (1) EVERY wave looks at the same 64 candidates (lane <-> candidate) and pulls against its own slice of the kept list
        //     (ranks pulled + wave, + 16, ...): a candidate's tests are spread over 16 threads.  Wave w publishes its verdict
        //     word; the AND is the set still alive.  The chunk's pairwise suppression ROWS are already there: they do not
        //     depend on what was kept, so they were computed while the previous chunk was being resolved (below) -- all of
        //     the chunk's IoUs off the serial chain of (2) (measured before the rows existed: 0.28 us per kept box in (2), 28
        //     of the kernel's 50 us), and since round 3 off the critical path altogether.
        bool alive = r < n_round;
        if (alive) alive = pull_against_kept(s_kcls, s_kbox, pulled + wave, kept_before, 16, jb, jc, true, thr);
        const uint64_t word = __ballot(alive);
        if (lane == 0) s_alive[wave] = word;
        __syncthreads();
      } else if (kStage == 2 && c0 < m_cov) {
        // rotated, first round, inside the matrix: the dead bits already hold the rows of every box kept so far; the chunk's
        // own rows are one word per candidate
        if (tid < kNmsChunk) {
          const uint32_t i = c0 + static_cast<uint32_t>(tid);
          s_sup[tid] = i < m_cov ? s_mat[i * (a.m_max / 64) + (c0 >> 6)] : 0ull;
        }
        __syncthreads();
      } else {
        // rotated: pull first (dead bits), then rows among the survivors only -- both through the pair queue
        if (tid < kNmsChunk) s_sup[tid] = 0;
        if (kept_before > pulled) pair_phase(false, c0, n_chunk, pulled, kept_before);
        else __syncthreads();
        pair_phase(true, c0, n_chunk, 0, 0);
      }
      cstamp(1);

      // (2) wave 0 resolves the chunk in rank order with scalar bit operations only: lane i holds row i, the first alive
      //     candidate is kept and clears its row's bits from the alive mask (v_readlane -> s_andn2).  No IoU, no LDS, one
      //     branch per kept box.
      if (wave == 0) {
        uint64_t word = ~0ull;
        if constexpr (NB == 4) {
          for (int w = 0; w < 16; ++w) word &= s_alive[w];
        } else {
          const uint64_t dead = static_cast<uint64_t>(s_dead[c0 >> 5]) | (static_cast<uint64_t>(s_dead[(c0 >> 5) + 1]) << 32);
          word = ~dead;
        }
        const uint64_t *rows = NB == 4 ? s_sup + ((c0 >> 6) & 1u) * kNmsChunk : s_sup;   // (axis-aligned: two buffers, in turn)
        const uint64_t my_row = static_cast<uint32_t>(lane) < n_chunk ? rows[lane] : 0;
        const uint32_t row_lo = static_cast<uint32_t>(my_row), row_hi = static_cast<uint32_t>(my_row >> 32);
        // (LDS values are VGPRs, "divergent" to the compiler: pin the loop state to SGPRs so the loop is scalar control flow)
        uint64_t mask = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(word >> 32))) << 32) |
                        static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(word)));
        uint64_t kept_mask = 0;
        int k_cnt = kept_before;
        while (mask && k_cnt < ndet) {
          // `hot`: alive candidates whose row hits something still alive.  The alive ones in front of the first of them are
          // kept as they stand -- they suppress nothing -- in ONE trip (random-init rotated heads: 64 of 64 kept took 64
          // trips of ~55 ns); a trip per hot candidate is what remains of the serial chain.
          const uint64_t hot = __ballot((my_row & mask) != 0) & mask;
          uint64_t run = hot ? mask & ((hot & (0ull - hot)) - 1ull) : mask;
          const int room = ndet - k_cnt;
          while (__popcll(run) > room) run &= ~(1ull << (63 - __clzll(static_cast<long long>(run))));   // (the list's last entries only)
          kept_mask |= run;
          k_cnt += __popcll(run);
          mask &= ~run;
          if (!hot || k_cnt >= ndet) break;
          const int l0 = __ffsll(static_cast<unsigned long long>(hot)) - 1;
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
          for (int k = 0; k < NB; ++k) s_kbox[my_rank * NB + k] = s_box[r * NB + k];
          s_kcls[my_rank] = s_cls[r];
          s_kscore[my_rank] = key_score(key);
          s_ksrc[my_rank] = static_cast<int32_t>(key_index(key));
        }
        if (lane == 0) {
          s_misc[34] = static_cast<uint32_t>(k_cnt);
          s_misc[38] = static_cast<uint32_t>(kept_mask);
          s_misc[39] = static_cast<uint32_t>(kept_mask >> 32);
        }
      } else if (NB == 4 && c0 + kNmsChunk < n_round) {
        // ... while the other 15 waves compute the NEXT chunk's rows into the other buffer (wasted only when this chunk
        // turns out to be the last one)
        chunk_rows(c0 + kNmsChunk, s_sup + (((c0 >> 6) + 1u) & 1u) * kNmsChunk, 1, 15);
      }
      __syncthreads();
      kept = __builtin_amdgcn_readfirstlane(static_cast<int>(s_misc[34]));
      if constexpr (kStage == 2) {
        if (c0 < m_cov) {
          // every box kept in this chunk ORs its row into the dead bits of the column blocks behind the chunk: thread
          // (candidate i, block w), one word each
          const uint32_t cov_blocks = (m_cov + 63) / 64;
          const uint32_t i = static_cast<uint32_t>(tid) >> 4, w = static_cast<uint32_t>(tid) & 15u;
          const uint64_t kept_mask = static_cast<uint64_t>(s_misc[38]) | (static_cast<uint64_t>(s_misc[39]) << 32);
          if (((kept_mask >> i) & 1ull) && w > (c0 >> 6) && w < cov_blocks) {
            const uint64_t word = s_mat[(c0 + i) * (a.m_max / 64) + w];
            if (static_cast<uint32_t>(word)) atomicOr(&s_dead[2 * w], static_cast<uint32_t>(word));
            if (static_cast<uint32_t>(word >> 32)) atomicOr(&s_dead[2 * w + 1], static_cast<uint32_t>(word >> 32));
          }
          __syncthreads();
        }
      }
      cstamp(2);
      cstamp(3);
    }
    if constexpr (kStage == 2) {
      if (first_step && kept < ndet) {                       // the first m_first candidates did not suffice: step 2 redoes the
        if (tid == 0) a.done[img] = 0;                       // round with the whole matrix (this launch writes no output)
        return;
      }
    }
    first_round = false;
    phase(3);
    last_round_kept = kept - kept_at_round_start;
    last_round_size = n_round;

    // ---- heavy suppression: drop everything the kept list already suppresses, in one parallel pass ----
    // Worth it only when everything that is left would be examined anyway: at this round's yield (kept per examined) the
    // boxes still missing need more candidates than remain.  Then the pass does no pair test the lazy rounds would not do
    // too, and saves their selection / sort / barrier overhead; otherwise it would test candidates nobody ever looks at.
    if (kept < ndet && left > 0 &&
        static_cast<unsigned long long>(ndet - kept) * n_round >= static_cast<unsigned long long>(left) * static_cast<uint32_t>(kept - kept_at_round_start)) {
      // source = the candidates no round has taken yet (sorted-run mode: the tails of the runs; generic: keys < upper);
      // survivors are appended to the key list (its front, in place) and the kernel continues in generic mode on them
      const uint32_t n_src = runs ? left : list_n;
      if (tid == 0) s_misc[37] = 0;
      __syncthreads();
      for (uint32_t f0 = 0; f0 < n_src; f0 += kNmsThreads) {   // block-uniform trip count
        const uint32_t f = f0 + tid;
        uint64_t key = 0;
        float cb[NB];
        float cc = 0.0f;
        bool have_box = false;
        if (f < n_src) {
          if (runs) {
            uint32_t q = 0, base = 0, acc = 0;                   // flat index f -> (run q, offset f - base) over the runs' tails
#pragma unroll
            for (uint32_t t = 0; t < 8; ++t) {                   // (selects only: a dynamically indexed private array would live in scratch)
              const uint32_t n_t = t < n_runs ? s_valid[t] - s_cursor[t] : 0u;
              const bool here = f >= acc && f - acc < n_t;
              q = here ? t : q;
              base = here ? acc : base;
              acc += n_t;
            }
            const uint32_t p = q * a.run_len + s_cursor[q] + (f - base);
            if constexpr (NB == 4) {                           // box and class with the score: ONE global latency, not two
#pragma unroll
              for (int k = 0; k < NB; ++k) cb[k] = in_b[static_cast<size_t>(p) * NB + k];
              cc = in_c[p];
              have_box = true;
            }
            key = make_key(in_s[p], p);
          } else {
            key = s_keys[f];
            key = key < upper ? key : 0;                       // already handed to a round
          }
        }
        bool alive = key != 0;
        if (alive && !have_box) {
          const uint32_t p = key_index(key);
#pragma unroll
          for (int k = 0; k < NB; ++k) cb[k] = in_b[static_cast<size_t>(p) * NB + k];
          cc = in_c[p];
        }
        if constexpr (NB == 4) {
          __syncthreads();                                     // (generic mode: this slab's keys are read before any is overwritten)
          if (alive) alive = pull_against_kept(s_kcls, s_kbox, 0, kept, 1, cb, cc, true, thr);
        } else {
          // stage the slab like a round and run the pull through the pair queue
          if (alive) {
#pragma unroll
            for (int k = 0; k < NB; ++k) s_box[tid * NB + k] = cb[k];
            s_cls[tid] = cc;
          }
          const uint64_t present = __ballot(alive);
          if (lane == 0) { s_dead[2 * wave] = ~static_cast<uint32_t>(present); s_dead[2 * wave + 1] = ~static_cast<uint32_t>(present >> 32); }
          __syncthreads();
          pair_phase(false, 0, kNmsThreads, 0, kept);
          alive = alive && !((s_dead[tid >> 5] >> (tid & 31)) & 1u);
        }
        const uint32_t slot = wave_append_slot(&s_misc[37], alive);
        if (alive) s_keys[slot] = key;
        __syncthreads();
      }
      list_n = __builtin_amdgcn_readfirstlane(s_misc[37]);
      left = list_n;
      upper = ~0ull;
      runs = false;
      pulled = kept;
      if (left > 0) key_range();
      phase(4);
    }
  }

  stamp(4);
  if (a.trace && tid == 0) {
    a.trace[(gridDim.x + blockIdx.x) * 8 + 5] = examined;
    a.trace[(gridDim.x + blockIdx.x) * 8 + 6] = K;
    a.trace[(gridDim.x + blockIdx.x) * 8 + 7] = static_cast<unsigned long long>(clock64() - shader_clock0);
  }
  // ---- outputs: kept boxes, then the zero-padded tail (box.py:322-324) ----
  if constexpr (kStage == 2) {
    if (a.step == 1 && tid == 0) a.done[img] = 1;
  }
  float *out_scores = reinterpret_cast<float *>(uniform_u64(s_out[0])), *out_boxes = reinterpret_cast<float *>(uniform_u64(s_out[1]));
  float *out_classes = reinterpret_cast<float *>(uniform_u64(s_out[2]));
  int32_t *out_indices = reinterpret_cast<int32_t *>(uniform_u64(s_out[3]));
  for (int t = tid; t < ndet; t += kNmsThreads) {
    const size_t o = static_cast<size_t>(img) * ndet + t;
    const bool v = t < kept;
    out_scores[o] = v ? s_kscore[t] : 0.0f;
    out_classes[o] = v ? s_kcls[t] : 0.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) out_boxes[o * NB + k] = v ? s_kbox[t * NB + k] : 0.0f;
    if (out_indices) out_indices[o] = v ? s_ksrc[t] : -1;
  }
}

}  // namespace odtk

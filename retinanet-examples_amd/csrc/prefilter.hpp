// prefilter.hpp -- kernel 1 of the decode path: one streaming pass over every score of every
// pyramid level of the whole batch; survivors of `score >= thresh` are compacted into a
// per-(level, image) candidate list as 64-bit (score, ~index) keys.
//
// Replaces reference steps D1-D3 (csrc/cuda/decode.cu:96-104: thrust::transform flags ->
// cub::DeviceSelect::Flagged -> cudaStreamSynchronize + D2H count) for ALL images and levels in
// one launch, with the count left on the device.  The templated forms additionally replace the
// three full passes the reference makes BEFORE its op (odtk/model.py:140 sigmoid, :160
// .contiguous() NHWC->NCHW copy, odtk/box.py:263 .float()): the kernel reads the head tensor as
// the convolution wrote it (bf16/fp16/fp32, NCHW or channels_last) and applies the sigmoid only
// to the few elements that can pass the threshold.
//
// Roofline: HBM-bound.  Algorithmic bytes = sizeof(T) per score, read once; writes are 8 B per
// survivor (<1 % of the reads at realistic densities).
//
// Work decomposition: a "tile" is kTile consecutive elements of one level's flat tensor; one
// workgroup (4 waves) per tile.
//   phase A (unrolled, branch-light): each lane issues all of its 16-byte loads, then builds a
//            64-bit hit mask with ONE compare per element in the raw domain (for logits: against a
//            conservative lower bound of the logit), takes block-local slots from an LDS atomic
//            and stages (raw bits, tile offset) pairs in LDS.
//   phase B (rolled, one call site): the staged entries get the exact test (sigmoid -> dtype
//            rounding -> `>= thresh`) and become keys.
//   copy-out: ONE wave reserves the global slots with ONE atomic per tile and writes the keys out
//            coalesced; the other three waves have already retired, so the ~1 us round trip of a
//            returning global atomic under streaming load never idles a whole workgroup.
//            (Per-candidate global atomics would serialise on one L2 word per image: ~88/us.)
// Tiles with more than kStageCap raw hits (saturated / adversarial inputs) and the <= 1 tile per
// image that straddles an image boundary take a rolled multi-round path over the same stage.
#pragma once

#include <type_traits>

#include "common.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

typedef uint32_t vuint4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;
constexpr int kTile = 16384;                     // elements per tile (64 per lane)
constexpr int kMaxSpanTiles = 2;                 // consecutive tiles per workgroup (ONE drain + ONE atomic for all):
                                                 // measured bs=8: 16-bit 57.4 -> 52.4 us with 2; fp32 is better at 1
constexpr int kStageCap = 2048;                  // hits staged in LDS per round (16 KiB; 4096 costs occupancy: 52 -> 62 us)
constexpr int kSubLists = 16;                    // candidate sub-lists (and counters) per segment

struct ScanLevel {
  const void *cls;       // level tensor, flat [batch * n] in its own layout
  uint64_t total;        // batch * n
  uint64_t cand_off;     // first key of this level's segment 0 in the candidate pool
  uint32_t n;            // scores per image = A*C*H*W
  uint32_t tile_begin;   // first workgroup of this level
  uint32_t seg_base;     // segment id of (level, image 0) = level * batch
  uint32_t cap;          // candidate capacity per SUB-LIST (kSubLists sub-lists per segment)
  uint32_t channels;     // A*C   (channels_last index mapping)
  uint32_t hw;           // H*W
  uint32_t channels_last;
  uint32_t tiles;        // tiles in this level = ceil(total / kTile)
  uint32_t chunk;        // tiles per interleave chunk = ceil(tiles / batch), a multiple of the span
  uint32_t pad_;
  const float *bias;     // kLogits + channels_last only: per-channel bias of the head's last conv (null: none)
};

struct ScanArgs {
  ScanLevel lv[ODTK_MAX_LEVELS];
  uint32_t *counts;      // [n_levels * batch][kSubLists] survivors per sub-list (exact, may exceed cap)
  uint64_t *cand;        // candidate pool
  int n_levels;
  int batch;
  int span;              // tiles per workgroup, 1..kMaxSpanTiles
  float thresh;          // threshold on the SCORE
  float raw_lo;          // kLogits: conservative lower bound on the raw logit of any survivor
};

// ---- element types ------------------------------------------------------------------------------
struct F32 { static constexpr int kPerLoad = 4; using storage = float; };
struct BF16 { static constexpr int kPerLoad = 8; using storage = uint16_t; };
struct F16 { static constexpr int kPerLoad = 8; using storage = uint16_t; };

__device__ __forceinline__ float bf16_bits_to_float(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float f16_bits_to_float(uint32_t h) {
  return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(h)));
}
__device__ __forceinline__ float round_to_bf16(float f) {   // round-to-nearest-even, as torch's cast
  uint32_t b = __float_as_uint(f);
  if ((b & 0x7fffffffu) > 0x7f800000u) return f;            // NaN
  b += 0x7fffu + ((b >> 16) & 1u);
  return __uint_as_float(b & 0xffff0000u);
}
__device__ __forceinline__ float round_to_f16(float f) { return static_cast<float>(static_cast<_Float16>(f)); }

// The score the op sees for a raw head value.  kLogits: torch's sigmoid formula
// (1 / (1 + exp(-x)) in fp32, ATen sigmoid_kernel_cuda) rounded to the tensor's own dtype, i.e.
// what `cls_head.sigmoid()` followed by `.float()` yields in the reference (model.py:140, box.py:263).
template <typename T, bool kLogits>
__device__ __forceinline__ float score_of(float raw) {
  if constexpr (!kLogits) {
    return raw;
  } else {
    const float s = 1.0f / (1.0f + expf(-raw));
    if constexpr (std::is_same_v<T, F32>) return s;
    else if constexpr (std::is_same_v<T, BF16>) return round_to_bf16(s);
    else return round_to_f16(s);
  }
}

template <typename T>
__device__ __forceinline__ float load_raw(const void *base, uint64_t idx) {
  if constexpr (std::is_same_v<T, F32>) {
    return static_cast<const float *>(base)[idx];
  } else {
    const uint32_t h = static_cast<const uint16_t *>(base)[idx];
    return std::is_same_v<T, BF16> ? bf16_bits_to_float(h) : f16_bits_to_float(h);
  }
}

// memory offset inside one image -> canonical flat NCHW index (the tie-break order, box.py:291-297)
__device__ __forceinline__ uint32_t canonical_index(uint32_t r, const ScanLevel &L) {
  if (!L.channels_last) return r;
  const uint32_t pix = r / L.channels, ch = r - pix * L.channels;
  return ch * L.hw + pix;
}
// canonical flat NCHW index -> memory offset inside one image
__device__ __forceinline__ uint32_t memory_offset(uint32_t i, uint32_t channels, uint32_t hw, uint32_t channels_last) {
  if (!channels_last) return i;
  const uint32_t ch = i / hw, pix = i - ch * hw;
  return pix * channels + ch;
}

// Per-channel raw-domain threshold in the tensor's own storage type, rounded TOWARDS -inf so that the
// stored value never exceeds the float threshold (the prefilter may only over-select).
template <typename T>
__device__ __forceinline__ typename T::storage threshold_to_storage(float t) {
  if constexpr (std::is_same_v<T, F32>) {
    return t;
  } else if constexpr (std::is_same_v<T, BF16>) {
    const uint32_t b = __float_as_uint(t);
    uint32_t h = b >> 16;                                  // truncation: towards zero
    if ((b & 0xffffu) && (b >> 31) && t == t) h += 1;      // negative with dropped bits: one step further down
    return static_cast<uint16_t>(h);
  } else {
    _Float16 h = static_cast<_Float16>(t);                 // nearest
    uint16_t bits = __builtin_bit_cast(uint16_t, h);
    if (static_cast<float>(h) > t) {                       // landed above: one representable step down
      if (bits == 0x0000u || bits == 0x8000u) bits = 0x8001u;
      else bits = (bits & 0x8000u) ? bits + 1 : bits - 1;
    }
    return bits;
  }
}
template <typename T>
__device__ __forceinline__ float storage_to_float(typename T::storage v) {
  if constexpr (std::is_same_v<T, F32>) return v;
  else if constexpr (std::is_same_v<T, BF16>) return bf16_bits_to_float(v);
  else return f16_bits_to_float(v);
}

// launch bounds: >= 8 waves/SIMD for the 16-bit forms (64 VGPRs, no spill), >= 6 for fp32 (80 VGPRs):
// more workgroups in their load phase while others drain (measured bf16 52.6 -> 48.4 us)
template <typename T, bool kLogits>
__global__ __launch_bounds__(kScanThreads, (sizeof(typename T::storage) == 2 ? 8 : 6)) void prefilter_scan_kernel(const ScanArgs a) {
  constexpr int kPer = T::kPerLoad;                        // elements per 16-byte load
  constexpr int kVec = kTile / (kScanThreads * kPer);      // loads per lane per tile: 16 (f32) or 8 (16-bit)
  __shared__ uint64_t s_stage[kStageCap];
  __shared__ uint32_t s_cnt;                               // raw hits staged this round
  __shared__ uint32_t s_ok;                                // of which pass the exact test
  // head bias folded in (ScanLevel::bias): per-channel raw-domain thresholds, A*C entries of T::storage
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  typename T::storage *s_thr = reinterpret_cast<typename T::storage *>(s_dyn);

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.lv[i].tile_begin) l = i;
  const ScanLevel &L = a.lv[l];

  const uint32_t kSpanTiles = static_cast<uint32_t>(a.span);
  // A workgroup owns a SPAN of `span` consecutive tiles.  Consecutive workgroups take spans from
  // DIFFERENT images (chunk c ~ image c of the flat level tensor): workgroups that run together
  // then reserve slots on `batch` different counters.  One counter word sustains only ~88 returning
  // atomics/us, and walking the tensor front to back keeps a single image's counter hot at a time.
  const uint32_t j = blockIdx.x - L.tile_begin;
  const uint32_t chunk_id = j % static_cast<uint32_t>(a.batch), in_chunk = j / static_cast<uint32_t>(a.batch);
  const uint32_t tile0 = chunk_id * L.chunk + in_chunk * kSpanTiles;
  if (tile0 >= L.tiles) return;                           // padding workgroup (block-uniform exit)
  const uint64_t span_base = static_cast<uint64_t>(tile0) * kTile;
  const uint32_t n = L.n;
  const float thr = a.thresh;
  const float raw_thr = kLogits ? a.raw_lo : a.thresh;
  const typename T::storage *span_ptr = static_cast<const typename T::storage *>(L.cls) + span_base;
  const uint64_t left = L.total - span_base;               // > 0 by construction
  const uint32_t span_len = left < static_cast<uint64_t>(kSpanTiles) * kTile ? static_cast<uint32_t>(left) : kSpanTiles * kTile;
  const uint32_t span_vec = span_len / kPer;               // whole 16-byte groups in this span

  if (tid == 0) { s_cnt = 0; s_ok = 0; }
  // With a bias the logit of element r is raw[r] + bias[r % channels] (channels_last, channels % kPer
  // == 0, checked by the host): "logit >= raw_lo" becomes "raw >= raw_lo - bias[c]" -- still ONE compare
  // per element, against a per-channel threshold that a lane fetches with one 16-byte LDS read per load.
  // (16-bit dtypes only -- the fp32 form has no registers to spare; the host rejects the combination)
  const float *bias = (kLogits && !std::is_same_v<T, F32>) ? L.bias : nullptr;   // block-uniform
  if (bias) {
    // (built before the tile's loads are issued: moving it behind them costs registers -- 44 B of scratch
    // and 49 -> 68 us on the common path, measured)
    for (uint32_t c = tid; c < L.channels; c += kScanThreads) {
      const float b = bias[c];
      s_thr[c] = threshold_to_storage<T>(raw_thr - b - (1e-3f + 1e-6f * fabsf(b)));   // margin >> fp32 rounding of raw + b
    }
  }
  auto logit_of = [&](float raw, uint64_t offset_in_level) -> float {   // exact-test input
    return bias ? raw + bias[static_cast<uint32_t>(offset_in_level % L.channels)] : raw;
  };

  // Every span appends to ONE of the segment's kSubLists sub-lists: ~15 k workgroups per launch all
  // want a slot reservation, so the returning atomics are spread over 16x more counter words
  // (measured at bs=8: fp32 118 -> 89 us, bf16 102 -> 56 us).
  const uint32_t sub = (tile0 / kSpanTiles) % kSubLists;
  auto counter_of = [&](uint32_t b) -> uint32_t * { return a.counts + (static_cast<size_t>(L.seg_base + b) * kSubLists + sub); };
  auto list_of = [&](uint32_t b) -> uint64_t * {
    return a.cand + L.cand_off + (static_cast<uint64_t>(b) * kSubLists + sub) * L.cap;
  };
  auto bits_to_raw = [](uint32_t bits) -> float {
    if constexpr (std::is_same_v<T, F32>) return __uint_as_float(bits);
    else if constexpr (std::is_same_v<T, BF16>) return bf16_bits_to_float(bits);
    else return f16_bits_to_float(bits);
  };

  // position of the span inside the level: image index and offset within the image
  const uint32_t b0 = static_cast<uint32_t>(span_base / n);
  const uint32_t r0 = static_cast<uint32_t>(span_base - static_cast<uint64_t>(b0) * n);
  const bool one_image = static_cast<uint64_t>(r0) + span_len <= n;

  // ---- phase A, once per tile of the span: loads, hit mask, stage (raw bits, span offset) in LDS ----
#pragma unroll 1
  for (uint32_t t = 0; t < kSpanTiles; ++t) {
    const uint32_t vec0 = t * (kTile / kPer);              // first 16-byte group of this tile
    if (vec0 >= span_vec && t > 0) break;
    const vuint4 *src = reinterpret_cast<const vuint4 *>(span_ptr) + vec0;
    const uint32_t n_vec = span_vec - vec0 < kTile / kPer ? span_vec - vec0 : kTile / kPer;

    // issue all loads first (kVec x 16 B per lane, lane-contiguous => fully coalesced)
    vuint4 v[kVec];
#pragma unroll
    for (int u = 0; u < kVec; ++u) {
      const uint32_t q = u * kScanThreads + tid;
      if (q < n_vec) v[u] = __builtin_nontemporal_load(src + q);     // streamed once: keep it out of L2's way
      else v[u] = std::is_same_v<T, F32> ? vuint4{0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u}
                                         : vuint4{0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u};   // NaNs
    }
    if (t == 0) __syncthreads();                           // s_cnt = 0 (and the threshold table) visible; overlaps the load latency

    // element e of load u  <->  tile element kPer*(u*256+tid)+e
    auto raw_at = [&](int u, int e) -> float {
      if constexpr (std::is_same_v<T, F32>) {
        return __uint_as_float(v[u][e]);
      } else {
        const uint32_t w = v[u][e >> 1];
        const uint32_t h = (e & 1) ? (w >> 16) : (w & 0xffffu);
        return std::is_same_v<T, BF16> ? bf16_bits_to_float(h) : f16_bits_to_float(h);
      }
    };
    // hit mask over the lane's 64 elements: bit (kPer*u + e); one compare each, NaN fails >=
    uint64_t mask = 0;
    if (!bias) {
#pragma unroll
      for (int u = 0; u < kVec; ++u) {
        uint32_t m = 0;
#pragma unroll
        for (int e = 0; e < kPer; ++e) m |= (raw_at(u, e) >= raw_thr ? 1u : 0u) << e;
        mask |= static_cast<uint64_t>(m) << (kPer * u);
      }
    } else {
      // channel group (kPer consecutive channels) of the lane's load u: (first group of the tile + u*256 + tid) mod G
      const uint32_t G = L.channels / kPer, step = kScanThreads % G;
      uint32_t g = static_cast<uint32_t>((span_base / kPer + vec0 + tid) % G);
#pragma unroll
      for (int u = 0; u < kVec; ++u) {
        const vuint4 tv = *reinterpret_cast<const vuint4 *>(s_dyn + static_cast<size_t>(g) * 16);
        uint32_t m = 0;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
          float te;
          if constexpr (std::is_same_v<T, F32>) te = __uint_as_float(tv[e]);
          else te = storage_to_float<T>(static_cast<uint16_t>((tv[e >> 1] >> (16 * (e & 1))) & 0xffffu));
          m |= (raw_at(u, e) >= te ? 1u : 0u) << e;
        }
        mask |= static_cast<uint64_t>(m) << (kPer * u);
        g += step;
        if (g >= G) g -= G;
      }
    }
    const uint32_t cnt = __popcll(mask);
    if (cnt) {
      uint32_t o = atomicAdd(&s_cnt, cnt);                 // block-local slots (order is irrelevant)
      if (o + cnt <= kStageCap) {
        const uint32_t elem0 = vec0 * kPer + kPer * tid;   // span offset of this lane's first element
        if constexpr (std::is_same_v<T, F32>) {
          // fp32: 16 loads are live (64 VGPRs) -- the fully unrolled walk is what fits in 80 registers
#pragma unroll
          for (int u = 0; u < kVec; ++u) {
            const uint32_t m = static_cast<uint32_t>(mask >> (kPer * u)) & ((1u << kPer) - 1u);
            if (m) {
#pragma unroll
              for (int e = 0; e < kPer; ++e)
                if (m & (1u << e))
                  s_stage[o++] = (static_cast<uint64_t>(v[u][e]) << 32) |
                                 static_cast<uint32_t>(elem0 + kPer * u * kScanThreads + e);
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < kVec; ++u) {
            uint32_t m = static_cast<uint32_t>(mask >> (kPer * u)) & ((1u << kPer) - 1u);
            // rolled: one trip per hit of this lane inside load u (the wave almost never needs a
            // second); an unrolled walk over the 64 bit positions costs ~2x the mask build itself
#pragma unroll 1
            while (m) {
              const uint32_t e = static_cast<uint32_t>(__builtin_ctz(m));
              m &= m - 1u;
              const uint32_t d = e >> 1;
              const uint32_t w = d == 0 ? v[u][0] : (d == 1 ? v[u][1] : (d == 2 ? v[u][2] : v[u][3]));
              s_stage[o++] = (static_cast<uint64_t>((w >> (16u * (e & 1u))) & 0xffffu) << 32) |
                             (elem0 + kPer * u * kScanThreads + e);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const uint32_t raw_tot = s_cnt;

  // ---- phase B + copy-out over `cnt_staged` staged (raw bits, span offset) entries ----
  auto drain = [&](uint32_t cnt_staged) {
    uint32_t ok_here = 0;
    for (uint32_t i = tid; i < cnt_staged; i += kScanThreads) {        // exact test, one call site
      const uint64_t ent = s_stage[i];
      const float s = score_of<T, kLogits>(logit_of(bits_to_raw(static_cast<uint32_t>(ent >> 32)),
                                                    span_base + static_cast<uint32_t>(ent)));
      uint64_t key = 0;
      if (!kLogits || s >= thr) {
        const uint32_t rr = r0 + static_cast<uint32_t>(ent);            // offset from image b0's start
        if (one_image) {
          key = make_key(s, canonical_index(rr, L));
          ++ok_here;
        } else {                                                        // boundary span: own atomics
          const uint32_t b = b0 + rr / n;
          const uint32_t slot = atomicAdd(counter_of(b), 1u);
          if (slot < L.cap) list_of(b)[slot] = make_key(s, canonical_index(rr % n, L));
        }
      }
      s_stage[i] = key;                                                 // 0 = dropped / already written
    }
    if (one_image) {
      const uint64_t bal = __ballot(ok_here != 0);
      if (bal) {                                                        // wave-uniform
        uint32_t w = ok_here;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) w += __shfl_xor(w, d, kWave);
        if (lane == 0) atomicAdd(&s_ok, w);
      }
    }
    __syncthreads();
    if (!one_image) return;
    const uint32_t ok_tot = s_ok;
    // ONE wave reserves the global slots (one returning atomic per span) and writes the keys out
    // coalesced; in the common single-round case waves 1..3 have nothing left to do and retire, so
    // the ~1 us round trip of the atomic under streaming load never idles the whole workgroup
    if (tid < kWave && ok_tot) {
      uint32_t base = 0;
      if (tid == 0) base = atomicAdd(counter_of(b0), ok_tot);
      base = __shfl(base, 0, kWave);
      uint64_t *dst = list_of(b0);
      uint32_t run = base;
      for (uint32_t i0 = 0; i0 < cnt_staged; i0 += kWave) {
        const uint64_t key = (i0 + lane < cnt_staged) ? s_stage[i0 + lane] : 0;
        const uint64_t m = __ballot(key != 0);
        const uint32_t pos = run + __popcll(m & ((1ull << lane) - 1ull));
        if (key != 0 && pos < L.cap) dst[pos] = key;
        run += __popcll(m);
      }
    }
  };

  if (raw_tot != 0 && raw_tot <= kStageCap) {
    drain(raw_tot);
  } else if (raw_tot > kStageCap) {
    // saturated span: re-walk it in rounds of kStageCap elements (4 per lane), rolled
    for (uint32_t c0 = 0; c0 < span_vec * kPer; c0 += kStageCap) {
      __syncthreads();
      if (tid == 0) { s_cnt = 0; s_ok = 0; }
      __syncthreads();
#pragma unroll 1
      for (int k = 0; k < kStageCap / kScanThreads; ++k) {
        const uint32_t e = c0 + k * kScanThreads + tid;
        bool hit = false;
        uint32_t bits = 0;
        if (e < span_vec * kPer) {
          const float raw = load_raw<T>(span_ptr, e);
          const float t_e = bias ? storage_to_float<T>(s_thr[static_cast<uint32_t>((span_base + e) % L.channels)]) : raw_thr;
          hit = raw >= t_e;
          if constexpr (std::is_same_v<T, F32>) bits = __float_as_uint(raw);
          else bits = static_cast<const uint16_t *>(static_cast<const void *>(span_ptr))[e];
        }
        // one LDS atomic per wave, not per hit: in a saturated span EVERY element is a hit, and 2048 atomics on one
        // LDS word per round serialise (this loop was 1.5 ms of the all-ones launch)
        const uint64_t m = __ballot(hit);
        if (m) {
          const int leader = __ffsll(static_cast<unsigned long long>(m)) - 1;
          uint32_t slot = 0;
          if (lane == leader) slot = atomicAdd(&s_cnt, static_cast<uint32_t>(__popcll(m)));
          slot = __shfl(slot, leader, kWave);
          if (hit) s_stage[slot + __popcll(m & ((1ull << lane) - 1ull))] = (static_cast<uint64_t>(bits) << 32) | e;
        }
      }
      __syncthreads();
      drain(s_cnt);
    }
  }

  // ---- scalar tail of the level (total % kPer elements, last span only) ----
  const uint32_t tail = span_len - span_vec * kPer;
  if (tail && static_cast<uint32_t>(tid) < tail) {
    const uint32_t toff = span_vec * kPer + tid;
    const float s = score_of<T, kLogits>(logit_of(load_raw<T>(span_ptr, toff), span_base + toff));
    if (s >= thr) {
      const uint64_t r = static_cast<uint64_t>(r0) + toff;
      const uint32_t b = b0 + static_cast<uint32_t>(r / n);
      const uint32_t slot = atomicAdd(counter_of(b), 1u);
      if (slot < L.cap) list_of(b)[slot] = make_key(s, canonical_index(static_cast<uint32_t>(r % n), L));
    }
  }
}

}  // namespace odtk

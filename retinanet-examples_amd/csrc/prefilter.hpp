// prefilter.hpp -- kernel 1 of the decode path: one streaming pass over every score of every
// pyramid level of the whole batch; survivors of `score >= thresh` are compacted into candidate
// lists of 64-bit (score, ~index) keys -- one fixed list per WAVE of a workgroup, no atomics.
//
// Replaces reference steps D1-D3 (csrc/cuda/decode.cu:96-104: thrust::transform flags ->
// cub::DeviceSelect::Flagged -> cudaStreamSynchronize + D2H count) for ALL images and levels in
// one launch, with the counts left on the device.  The templated forms additionally replace the
// three full passes the reference makes BEFORE its op (odtk/model.py:140 sigmoid, :160
// .contiguous() NHWC->NCHW copy, odtk/box.py:263 .float()): the kernel reads the head tensor as
// the convolution wrote it (bf16/fp16/fp32, NCHW or channels_last) and applies the sigmoid only
// to the few elements that can pass the threshold.
//
// Roofline: HBM-bound.  Algorithmic bytes = sizeof(T) per score, read once; writes are 8 B per
// survivor (<1 % of the reads at realistic densities).
//
// Work decomposition (round 4).  A "span" is 1-2 tiles of 16 384 consecutive elements of ONE image
// (spans never straddle images); one 256-thread workgroup per span; inside it every wave is on its own:
//   scan  : a lane issues all of its 16-byte loads, then ONE v_cmp per element whose result is the
//           64-lane mask in a scalar register pair (wave64: the compare IS the ballot).  The masks of a
//           load are OR-ed on the scalar unit; only when one is non-zero (scalar branch) do the hit
//           lanes stage (raw bits, span offset) in the wave's own LDS region, at a cursor that lives in
//           an SGPR -- no LDS atomic, no per-lane hit mask, ~12 VALU instructions per 16 bytes.
//           bf16: the high half of a dword is compared WITHOUT unpacking -- the dword itself, read as a
//           float, is the element plus < 1 bf16 ulp of garbage mantissa, and the threshold's low half is
//           set to 0xffff for negative thresholds so that the comparison of the magnitudes comes out
//           exactly as for the bare elements; `!(w < t)` instead of `w >= t` lets NaN / inf-with-garbage
//           through to the exact test (the prefilter may over-select, never under-select).
//   drain : the wave applies the exact test (sigmoid -> dtype rounding -> `>= thresh`) to its staged
//           entries and writes the keys, compacted with a ballot prefix, to ITS OWN sub-list of the span's
//           region in the candidate pool, then the sub-list's length.  No reservation, no returning
//           atomic (~1 us under streaming load), no workgroup barrier: the four waves retire
//           independently.  A wave with more than kWaveStage raw hits writes kListOverflow instead and
//           the consumer (select_decode.hpp) re-reads that span's raw scores: capacity is a speed knob,
//           never a result.
// Round 1-3 reserved list space with one returning global atomic per span (16 sub-lists per segment to
// spread them) and let one wave of four write out; the atomics, the interleaving they forced on the
// tile order and the separate histogram / filter launches that consumed the lists are gone.
#pragma once

#include <type_traits>

#include "common.hpp"
#include "fastdiv.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

typedef uint32_t vuint4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;
constexpr int kScanWaves = kScanThreads / kWave;
constexpr int kTile = 16384;                     // elements per tile (64 per lane)
constexpr int kMaxSpanTiles = 4;                 // most tiles per workgroup (a 16-bit staged offset); the launch uses 2 for 16-bit inputs, 1 for fp32 (measured)
constexpr int kWaveStage = 512;                  // raw hits a wave stages in LDS = keys its sub-list holds (6.25 % of its elements)
constexpr int kSpanCap = kScanWaves * kWaveStage;   // keys of a span's region in the candidate pool
constexpr uint32_t kListOverflow = 0xffffffffu;  // sub-list length: "more than kWaveStage raw hits, read the span's raw scores"

// Per-(level, image) segment state of select_decode's multi-workgroup route; zeroed by the prefilter (workgroup of span 0).
constexpr uint32_t kSelMaxParts = 64;   // workgroups per segment at most (= select_decode.hpp kMaxParts)
struct SelSeg {
  uint32_t surv_count;     // keys appended to the segment's survivor list (tournament route)
  uint32_t arrived;        // workgroups whose slice histogram is in `hist` (first ticket: the segment-local barrier counts on it)
  uint32_t route;          // 0 undecided | kRouteCoop | kRouteTournament: ONE compare-and-swap decides for the whole segment
  uint32_t arrived2;       // workgroups that have published their keys (second ticket: the last one finishes the segment)
  uint32_t hist[1 << 11];  // histogram (2048 equal bins of the key range, reversed) of the segment's keys
  uint32_t run_len[kSelMaxParts];   // cooperative route: keys in workgroup g's published list (its keys >= the segment's threshold)
};
constexpr uint32_t kRouteCoop = 1, kRouteTournament = 2;

struct ScanLevel {
  const void *cls;       // level tensor, [batch][n] in its own layout
  uint64_t key_off;      // first key of this level's span regions in the candidate pool
  uint32_t cnt_off;      // first sub-list length of this level in `counts`
  uint32_t n;            // scores per image = A*C*H*W
  uint32_t blk_begin;    // first workgroup of this level
  uint32_t spans;        // spans per image = ceil(n / span elements)
  uint32_t seg_base;     // segment id of (level, image 0) = level * batch
  uint32_t channels;     // A*C   (channels_last index mapping)
  uint32_t hw;           // H*W
  uint32_t channels_last;
  FastDiv by_channels;
  const float *bias;     // kLogits + channels_last only: per-channel bias of the head's last conv (null: none)
  const float *table;    // with bias: the threshold table prepared by odtk_prefilter_thresholds, or null
};

struct ScanArgs {
  ScanLevel lv[ODTK_MAX_LEVELS];
  uint32_t *counts;      // sub-list lengths: [level][image][span][kScanWaves]
  uint64_t *cand;        // candidate pool: per span kSpanCap keys, kWaveStage per wave
  SelSeg *sel;           // [n_levels * batch]
  int n_levels;
  int batch;
  int span;              // tiles per workgroup, 1..kMaxSpanTiles
  int image_major;       // workgroup order inside a level: 1 = image after image (one front), 0 = images interleaved
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

// canonical flat NCHW index -> memory offset inside one image
__device__ __forceinline__ uint32_t memory_offset(uint32_t i, uint32_t channels, uint32_t hw, uint32_t channels_last) {
  if (!channels_last) return i;
  const uint32_t ch = i / hw, pix = i - ch * hw;
  return pix * channels + ch;
}

// A raw-domain threshold in the tensor's own storage type, rounded TOWARDS -inf so that the stored value
// never exceeds the float threshold (the prefilter may only over-select).
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

// The pair of float thresholds the scan compares the two halves of a dword against, for a threshold t (float):
//   lo : for the element in the LOW half, unpacked exactly (`x >= lo`);
//   hi : for the element in the HIGH half.  fp16: unpacked exactly too, same value.  bf16: compared as the whole dword
//        read as a float, `!(w < hi)`: w = element with up to 0xffff of garbage below its mantissa, so for a negative
//        threshold the low half of `hi` is 0xffff (then |w| <= |hi|  <=>  |x| <= |t| on the 16-bit magnitudes) and
//        +0 becomes the smallest-magnitude negative pattern (so that a -0.0 element with garbage still passes).
template <typename T>
__device__ __forceinline__ void scan_thresholds(float t, float *lo, float *hi) {
  const typename T::storage s = threshold_to_storage<T>(t);
  const float f = storage_to_float<T>(s);
  *lo = f;
  if constexpr (std::is_same_v<T, BF16>) {
    uint32_t h = static_cast<uint32_t>(s);
    if (h == 0u) h = 0x8000u;
    *hi = __uint_as_float((h << 16) | ((h & 0x8000u) ? 0xffffu : 0u));
  } else {
    *hi = f;
  }
}

// The threshold table of a level: 8 header words (word 0: a key of what the table was made for -- raw-domain threshold, dtype,
// channel count --, the rest padding) then, per group of 8 consecutive channels, the 8 floats a lane compares one 16-byte load against: [0..3] for the
// low halves of its four dwords (even channels), [4..7] for the high halves (odd channels).
constexpr uint32_t kTableMagic = 0x4f44544bu;    // "ODTK"
constexpr int kTableHeader = 8;
__host__ __device__ inline uint32_t table_key(uint32_t raw_thr_bits, uint32_t dtype, uint32_t channels) {
  return kTableMagic ^ raw_thr_bits ^ (dtype << 28) ^ (channels * 0x9e3779b1u);
}

template <typename T>
__device__ __forceinline__ void table_entry(float raw_thr, float bias_c, uint32_t c, float *body) {
  float lo, hi;
  scan_thresholds<T>(raw_thr - bias_c - (1e-3f + 1e-6f * fabsf(bias_c)), &lo, &hi);   // margin >> fp32 rounding of raw + b
  const uint32_t e = c & 7u;
  body[(c & ~7u) + (e >> 1) + ((e & 1u) ? 4u : 0u)] = (e & 1u) ? hi : lo;
}

template <typename T>
__global__ __launch_bounds__(256) void prefilter_table_kernel(const float *__restrict__ bias, uint32_t channels, float raw_thr,
                                                              uint32_t dtype, float *__restrict__ table) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c < channels) table_entry<T>(raw_thr, bias[c], c, table + kTableHeader);
  if (c < static_cast<uint32_t>(kTableHeader)) table[c] = __uint_as_float(c == 0 ? table_key(__float_as_uint(raw_thr), dtype, channels) : 0u);
}

// launch bounds: >= 8 waves/SIMD for the 16-bit forms (64 VGPRs), >= 6 for fp32 (80 VGPRs): more workgroups in their
// load phase while others drain (measured bf16 52.6 -> 48.4 us).  The element-load form (odd shapes) is not held to it.
template <typename T, bool kLogits, bool kAligned>
__global__ __launch_bounds__(kScanThreads, (!kAligned ? 2 : sizeof(typename T::storage) == 2 ? 8 : 6)) void prefilter_scan_kernel(const ScanArgs a) {
  constexpr bool k16 = sizeof(typename T::storage) == 2;
  constexpr int kPer = T::kPerLoad;                        // elements per 16-byte load
  constexpr int kVec = kTile / (kScanThreads * kPer);      // loads per lane per tile: 16 (f32) or 8 (16-bit)
  // staged hit: 16-bit types (raw bits << 16) | span offset (a span has <= 2^16 elements) in 4 bytes; fp32 (bits << 32) | offset
  using stage_t = std::conditional_t<k16, uint32_t, uint64_t>;
  __shared__ stage_t s_stage[kSpanCap];                    // [wave][kWaveStage]: every wave stages and drains its own region
  // head bias folded in (ScanLevel::bias): per-channel float thresholds, 8 per group of 8 consecutive channels: [0..3] for
  // the low halves of the load's four dwords (even channels), [4..7] for the high halves (odd channels)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
  float *s_thr = reinterpret_cast<float *>(s_dyn);

  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.lv[i].blk_begin) l = i;
  const ScanLevel &L = a.lv[l];

  // workgroup -> (image, span).  image_major: the launch sweeps one image after the other, front to back; otherwise
  // consecutive workgroups take the same span of different images (A/B knob)
  const uint32_t j = blockIdx.x - L.blk_begin;
  const uint32_t batch = static_cast<uint32_t>(a.batch);
  const uint32_t b = a.image_major ? j / L.spans : j % batch;
  const uint32_t s = a.image_major ? j - b * L.spans : j / batch;
  const uint32_t span_elems = static_cast<uint32_t>(a.span) * kTile;
  const uint32_t n = L.n;
  const uint32_t r0 = s * span_elems;                      // offset of the span inside its image
  const uint32_t span_len = n - r0 < span_elems ? n - r0 : span_elems;
  const float thr = a.thresh;
  const float raw_thr = kLogits ? a.raw_lo : a.thresh;
  const typename T::storage *span_ptr = static_cast<const typename T::storage *>(L.cls) + (static_cast<uint64_t>(b) * n + r0);

  if (s == 0) {                                            // this segment's select_decode state starts at zero
    SelSeg *S = a.sel + (L.seg_base + b);
    uint4 *h = reinterpret_cast<uint4 *>(S->hist) + 2 * tid;   // 2048 words = 256 threads x 2 x 16 bytes
    h[0] = make_uint4(0u, 0u, 0u, 0u);
    h[1] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) { S->surv_count = 0; S->arrived = 0; S->route = 0; S->arrived2 = 0; }
  }

  // With a bias the logit of element r is raw[r] + bias[r % channels] (channels_last, channels % kPer == 0, checked by
  // the host): "logit >= raw_lo" becomes "raw >= raw_lo - bias[c]" -- still ONE compare per element, against a
  // per-channel threshold that a lane fetches with two 16-byte LDS reads per load.
  // (16-bit dtypes only; the host rejects the fp32 combination)
  const float *bias = (kLogits && k16) ? L.bias : nullptr;   // block-uniform
  // The table comes (a) ready-made from the caller (up to 1024 channels): its header and one 16-byte load per lane are issued
  // IN FRONT of the tile's loads, checked and written to LDS while those are in flight --, or (b) is built here from the
  // bias, before the tile's loads are issued (memory operations return in order: a bias load issued behind them would wait
  // for the whole tile -- measured in round 1, and again as 8 B of scratch at the 64-register bound in round 4).
  const bool tbl_pending = bias && L.table && L.channels <= 4u * kScanThreads;   // (block-uniform: kernel arguments only)
  if (bias && !tbl_pending)
    for (uint32_t c = tid; c < L.channels; c += kScanThreads) table_entry<T>(raw_thr, bias[c], c, s_thr);
  float u_lo = raw_thr, u_hi = raw_thr;                    // without a bias: one threshold pair for every channel
  if constexpr (k16) scan_thresholds<T>(raw_thr, &u_lo, &u_hi);

  stage_t *my_stage = s_stage + wave * kWaveStage;
  uint32_t cursor = 0;                                     // raw hits of this wave so far (wave-uniform: lives in an SGPR)

  // ---- scan, once per tile of the span ----
#pragma unroll 1
  for (uint32_t t = 0; t < static_cast<uint32_t>(a.span); ++t) {
    const uint32_t tile_off = t * kTile;
    if (tile_off >= span_len) break;
    const uint32_t tile_len = span_len - tile_off < static_cast<uint32_t>(kTile) ? span_len - tile_off : kTile;

    // (a ready-made table: header + this lane's 16 bytes, issued in front of the tile's loads -- inside the iteration, so that
    // the wait-count pass orders them against the tile's loads instead of draining everything at the loop's entry)
    vuint4 tbl = vuint4{0u, 0u, 0u, 0u};
    uint32_t hdr = 0;
    if (t == 0 && tbl_pending) {
      hdr = *reinterpret_cast<const uint32_t *>(L.table);  // what the table was made for: checked once it has arrived
      if (static_cast<uint32_t>(tid) < L.channels / 4) tbl = *(reinterpret_cast<const vuint4 *>(L.table + kTableHeader) + tid);
    }
    // issue all loads first (kVec x 16 B per lane, lane-contiguous => fully coalesced).  Padding: a large negative
    constexpr uint32_t kPadWord = k16 ? 0xff7fff7fu : 0xff7fffffu;
    vuint4 v[kVec];
    if constexpr (kAligned) {
      const vuint4 *src = reinterpret_cast<const vuint4 *>(span_ptr + tile_off);
      if (tile_len == static_cast<uint32_t>(kTile)) {      // (block-uniform) a whole tile: nothing to predicate
#pragma unroll
        for (int u = 0; u < kVec; ++u) v[u] = __builtin_nontemporal_load(src + (u * kScanThreads + tid));   // streamed once: keep it out of L2's way
      } else {
        const uint32_t n_vec = tile_len / kPer;            // n % kPer == 0: whole vectors only
#pragma unroll
        for (int u = 0; u < kVec; ++u) {
          const uint32_t q = u * kScanThreads + tid;
          if (q < n_vec) v[u] = __builtin_nontemporal_load(src + q);
          else v[u] = vuint4{kPadWord, kPadWord, kPadWord, kPadWord};
        }
      }
    } else {
      // images that do not start on a 16-byte boundary (n % kPer != 0: odd test shapes): element loads, same register layout
      const typename T::storage *src = span_ptr + tile_off;
#pragma unroll
      for (int u = 0; u < kVec; ++u) {
        const uint32_t e0 = (u * kScanThreads + tid) * kPer;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if constexpr (k16) {
            const uint32_t i0 = e0 + 2 * d, i1 = i0 + 1;
            const uint32_t lo = i0 < tile_len ? static_cast<uint32_t>(src[i0]) : (kPadWord & 0xffffu);
            const uint32_t hi = i1 < tile_len ? static_cast<uint32_t>(src[i1]) : (kPadWord >> 16);
            v[u][d] = lo | (hi << 16);
          } else {
            v[u][d] = e0 + d < tile_len ? __float_as_uint(src[e0 + d]) : kPadWord;
          }
        }
      }
    }
    if (t == 0) {
      if (tbl_pending) {
        const bool ready = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(hdr)) ==
                           table_key(__float_as_uint(raw_thr), std::is_same_v<T, BF16> ? ODTK_BF16 : ODTK_F16, L.channels);
        // A table made for something else (another threshold / dtype / channel count: the caller's doing) is replaced by
        // thresholds that EVERY element passes (-inf; NaN for bf16's `!(w < t)` form): the waves overflow their stage, mark
        // their sub-lists kListOverflow and select_decode re-reads the raw scores -- slow, and exact.
        if (!ready) {
          const uint32_t pass = (std::is_same_v<T, BF16> && (tid & 1)) ? 0x7fc00000u : 0xff800000u;   // (odd vectors: the high halves)
          tbl = vuint4{pass, pass, pass, pass};
        }
        if (static_cast<uint32_t>(tid) < L.channels / 4) *(reinterpret_cast<vuint4 *>(s_thr) + tid) = tbl;
      }
      __syncthreads();                                     // the threshold table is visible; overlaps the load latency
    }

    // channel group (kPer consecutive channels) of the lane's load u: (first group of the tile + u*256 + tid) mod G
    uint32_t g = 0, g_step = 0, G = 1;
    if (bias) {
      G = L.channels / kPer;
      g_step = kScanThreads % G;
      g = ((r0 + tile_off) / kPer + tid) % G;
    }
    const uint32_t lane_off = tile_off + kPer * tid;       // span offset of the lane's first element of load 0

#pragma unroll
    for (int u = 0; u < kVec; ++u) {
      // thresholds of this load's elements: tl[d] for the low half / the fp32 word d, th[d] for the high half of dword d
      float tl[4], th[4];
      if (bias) {
        const vuint4 a0 = *reinterpret_cast<const vuint4 *>(s_dyn + static_cast<size_t>(g) * 32);
        const vuint4 a1 = *reinterpret_cast<const vuint4 *>(s_dyn + static_cast<size_t>(g) * 32 + 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) { tl[d] = __uint_as_float(a0[d]); th[d] = __uint_as_float(a1[d]); }
        g += g_step;
        if (g >= G) g -= G;
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) { tl[d] = u_lo; th[d] = u_hi; }
      }
      // one compare per element; on wave64 its result IS the 64-lane mask (an SGPR pair)
      bool hit[kPer];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t w = v[u][d];
        if constexpr (std::is_same_v<T, F32>) {
          hit[d] = __uint_as_float(w) >= tl[d];
        } else if constexpr (std::is_same_v<T, BF16>) {
          hit[2 * d] = __uint_as_float(w << 16) >= tl[d];
          hit[2 * d + 1] = !(__uint_as_float(w) < th[d]);
        } else {
          hit[2 * d] = f16_bits_to_float(w & 0xffffu) >= tl[d];
          hit[2 * d + 1] = f16_bits_to_float(w >> 16) >= th[d];
        }
      }
      uint64_t m[kPer], any = 0;
#pragma unroll
      for (int e = 0; e < kPer; ++e) { m[e] = __ballot(hit[e]); any |= m[e]; }
      if (any) {                                           // scalar branch: no lane of the wave has a hit in most loads
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
          if (m[e]) {                                      // scalar
            const uint32_t cnt = static_cast<uint32_t>(__popcll(m[e]));
            if (cursor + cnt <= static_cast<uint32_t>(kWaveStage) && hit[e]) {
              const uint32_t pos = cursor + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m[e] >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m[e]), 0u));
              const uint32_t off = lane_off + static_cast<uint32_t>(kPer * u * kScanThreads + e);
              if constexpr (k16) {
                const uint32_t w = v[u][e >> 1];
                const uint32_t bits = (e & 1) ? (w >> 16) : (w & 0xffffu);
                my_stage[pos] = (bits << 16) | off;
              } else {
                my_stage[pos] = (static_cast<uint64_t>(v[u][e]) << 32) | off;
              }
            }
            cursor += cnt;                                 // beyond kWaveStage: overflow, nothing more is staged
          }
        }
      }
    }
  }

  // ---- drain: exact test + compaction, this wave's own entries into this wave's own sub-list ----
  const uint32_t list = L.cnt_off + (b * L.spans + s) * kScanWaves + wave;
  uint64_t *dst = a.cand + L.key_off + (static_cast<uint64_t>(b) * L.spans + s) * kSpanCap + wave * kWaveStage;
  const bool overflow = cursor > static_cast<uint32_t>(kWaveStage);
  uint32_t run = 0;
  if (cursor != 0 && !overflow) {
    // the entries were written by other lanes of this wave: LDS operations of a wave complete in order, the fence keeps
    // the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (uint32_t i0 = 0; i0 < cursor; i0 += kWave) {      // (wave-uniform trip count: the ballot is legal)
      const uint32_t i = i0 + lane;
      bool ok = false;
      uint64_t key = 0;
      if (i < cursor) {
        uint32_t off;
        float raw;
        if constexpr (k16) {
          const uint32_t ent = my_stage[i];
          off = ent & 0xffffu;
          raw = storage_to_float<T>(static_cast<uint16_t>(ent >> 16));
        } else {
          const uint64_t ent = my_stage[i];
          off = static_cast<uint32_t>(ent);
          raw = __uint_as_float(static_cast<uint32_t>(ent >> 32));
        }
        if (off < span_len) {                              // (padding lanes can pass a threshold of -inf)
          const uint32_t rr = r0 + off;                    // offset inside the image
          uint32_t index = rr;                             // canonical flat NCHW index (the tie-break order, box.py:291-297)
          float x = raw;
          if (L.channels_last) {
            uint32_t ch;
            const uint32_t pix = fastdivmod(rr, L.by_channels, &ch);
            index = ch * L.hw + pix;
            if (bias) x = raw + bias[ch];
          }
          const float sc = score_of<T, kLogits>(x);
          ok = sc >= thr;
          key = make_key(sc, index);
        }
      }
      const uint64_t mk = __ballot(ok);
      if (ok) dst[run + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mk >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mk), 0u))] = key;
      run += static_cast<uint32_t>(__popcll(mk));
    }
  }
  if (lane == 0) a.counts[list] = overflow ? kListOverflow : run;
}

}  // namespace odtk

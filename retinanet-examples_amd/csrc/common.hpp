// common.hpp -- shared device helpers for the gfx950 post-processing kernels.
//
// Target: MI355X / gfx950 only (wave64, 256 CUs in 8 XCDs, 160 KiB LDS per CU).
// Parity rule: every arithmetic expression that feeds an output value is written in the
// exact operation order of the reference's CPU path (odtk/box.py) and the library is built
// with -ffp-contract=off so hipcc never fuses a*b+c into an FMA the CPU does not perform.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace odtk {

constexpr int kWave = 64;

// ---------------------------------------------------------------------------------------
// 64-bit selection key:  (orderable(score) << 32) | ~flat_index
// Larger key == better candidate: score descending, then flat NCHW index ascending -- the
// canonical tie rule (stable descending sort of an index-ascending list; box.py:289 with
// torch.sort(stable=True), equal to the CUDA path's stable radix sort decode.cu:111-112).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_to_ordered(float s) {
  s = s + 0.0f;  // -0.0 -> +0.0 so both zeros compare equal, as torch.sort does
  uint32_t b = __float_as_uint(s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t o) {
  uint32_t b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(b);
}
__device__ __forceinline__ uint64_t make_key(float score, uint32_t index) {
  return (static_cast<uint64_t>(float_to_ordered(score)) << 32) | static_cast<uint32_t>(~index);
}
__device__ __forceinline__ float key_score(uint64_t k) { return ordered_to_float(static_cast<uint32_t>(k >> 32)); }
__device__ __forceinline__ uint32_t key_index(uint64_t k) { return ~static_cast<uint32_t>(k); }

// ---------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t n = __shfl_up(v, d, kWave);
    if (lane >= d) v += n;
  }
  return v;
}

// The same on the DPP network (row shifts + row broadcasts: six VALU instructions, no LDS crossbar round trip per step as
// ds_bpermute-based shuffles pay): the sequence LLVM's atomic optimizer emits for gfx9 wave64 scans.
__device__ __forceinline__ uint32_t wave_inclusive_sum_dpp(uint32_t v) {
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x111, 0xf, 0xf, false));   // row_shr:1
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x112, 0xf, 0xf, false));   // row_shr:2
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x114, 0xf, 0xf, false));   // row_shr:4
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x118, 0xf, 0xf, false));   // row_shr:8
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
  v += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
  return v;
}
// maximum of a 64-bit value over the wave (every lane gets it): inclusive max-scan on the DPP network, lane 63 holds the result
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#define ODTK_DPP_MAX_STEP(ctrl, rows)                                                                                         \
  {                                                                                                                           \
    const uint32_t lo_ = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v)), ctrl, rows, 0xf, false)); \
    const uint32_t hi_ = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<uint32_t>(v >> 32)), ctrl, rows, 0xf, false)); \
    const uint64_t p_ = (static_cast<uint64_t>(hi_) << 32) | lo_;                                                             \
    v = p_ > v ? p_ : v;                                                                                                      \
  }
  ODTK_DPP_MAX_STEP(0x111, 0xf) ODTK_DPP_MAX_STEP(0x112, 0xf) ODTK_DPP_MAX_STEP(0x114, 0xf) ODTK_DPP_MAX_STEP(0x118, 0xf)
  ODTK_DPP_MAX_STEP(0x142, 0xa) ODTK_DPP_MAX_STEP(0x143, 0xc)
#undef ODTK_DPP_MAX_STEP
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), 63);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), 63);
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// Append slot for the lanes with `pred` in a list whose cursor is `counter` (LDS or global): ONE atomic per wave, issued by
// the first participating lane, instead of one per element -- atomics on a single word serialise (1024 LDS atomics on
// one cursor cost ~9 us).  Every lane of the wave must call it (ballot inside); the result is meaningful where pred holds.
__device__ __forceinline__ uint32_t wave_append_slot(uint32_t *counter, bool pred) {
  const uint64_t m = __ballot(pred);
  if (!m) return 0;                                        // wave-uniform
  const int lane = lane_id();
  const int leader = __ffsll(static_cast<unsigned long long>(m)) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, static_cast<uint32_t>(__popcll(m)));
  base = __shfl(base, leader, kWave);
  return base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
}

// torch.min / torch.max propagate NaN (box.py:107 `torch.max(m, torch.min(t, M))`)
__device__ __forceinline__ float clamp_like_torch(float t, float hi) {
  float mn = (t != t) ? t : (t < hi ? t : hi);
  return (mn != mn) ? mn : (mn > 0.0f ? mn : 0.0f);
}

// torch.max / torch.min propagate NaN
__device__ __forceinline__ float tmax_nan(float a, float b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ float tmin_nan(float a, float b) { return (a < b || a != a) ? a : b; }

// Correctly rounded fp32 exp (double-precision exp rounded once).  torch's CPU exp is within
// 1 ulp of this (measured: 1.1 % of inputs differ, by exactly 1 ulp) -- see DESIGN.md "exp".
__device__ __forceinline__ float exp_cr(float x) { return static_cast<float>(exp(static_cast<double>(x))); }

// A value that is the same in every lane, moved to scalar registers.  Values read from LDS (or derived from them) are
// "divergent" to the compiler and live in VGPRs even when every lane holds the same number; block-wide state that stays
// alive across register-hungry code (the polygon clip) belongs in SGPRs.
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p) { return reinterpret_cast<T *>(uniform_u64(reinterpret_cast<uint64_t>(p))); }

}  // namespace odtk

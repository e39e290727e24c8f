// prefilter.hpp -- kernel 1 of the decode path: one streaming pass over every score of every
// pyramid level of the whole batch; survivors of `score >= thresh` are compacted into a
// per-(level, image) candidate list as 64-bit (score, ~index) keys.
//
// Replaces reference steps D1-D3 (csrc/cuda/decode.cu:96-104: thrust::transform flags ->
// cub::DeviceSelect::Flagged -> cudaStreamSynchronize + D2H count) for ALL images and levels in
// one launch, with the count left on the device.
//
// Roofline: HBM-bound.  Algorithmic bytes = 4 B per score (read once); writes are 8 B per
// survivor (<1 % of the reads at realistic densities).
//
// Work decomposition: a "tile" is kTile consecutive elements of one level's flat
// [batch * A*C*H*W] tensor; one workgroup (4 waves) per tile, each lane issues kVec independent
// 16-byte loads before it touches any of them (64 KiB in flight per workgroup).  Survivors are
// counted per lane, block-scanned, and the workgroup reserves its slots with ONE global atomic
// per tile (per-candidate atomics would serialise on one L2 word per image: ~88 atomics/us).
#pragma once

#include "common.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

typedef float vfloat4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;
constexpr int kVec = 16;                                  // float4 loads per lane per tile
constexpr int kTile = kScanThreads * kVec * 4;            // 16384 scores = 64 KiB per workgroup

struct ScanLevel {
  const void *cls;       // level tensor, flat [batch * n]
  uint64_t total;        // batch * n
  uint64_t cand_off;     // first key of this level's segment 0 in the candidate pool
  uint32_t n;            // scores per image = A*C*H*W
  uint32_t tile_begin;   // first workgroup of this level
  uint32_t seg_base;     // segment id of (level, image 0) = level * batch
  uint32_t cap;          // candidate capacity per segment
};

struct ScanArgs {
  ScanLevel lv[ODTK_MAX_LEVELS];
  uint32_t *counts;      // [n_levels * batch] survivors per segment (exact, may exceed cap)
  uint64_t *cand;        // candidate pool
  int n_levels;
  int batch;
  float thresh;
};

__global__ __launch_bounds__(kScanThreads) void prefilter_scan_kernel(const ScanArgs a) {
  __shared__ uint32_t s_wave_tot[kScanThreads / kWave];
  __shared__ uint32_t s_base;

  const int tid = threadIdx.x;
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.lv[i].tile_begin) l = i;
  const ScanLevel &L = a.lv[l];

  const uint64_t tile_base = static_cast<uint64_t>(blockIdx.x - L.tile_begin) * kTile;
  const uint64_t total = L.total;
  const uint32_t n = L.n;
  const float thr = a.thresh;
  const vfloat4 *src = reinterpret_cast<const vfloat4 *>(static_cast<const float *>(L.cls) + tile_base);
  const uint64_t left = total - tile_base;                 // > 0 by construction
  const uint32_t tile_len = left < kTile ? static_cast<uint32_t>(left) : kTile;
  const uint32_t n_vec = tile_len >> 2;                    // whole float4s in this tile

  // ---- issue all loads first (kVec x 16 B per lane, fully coalesced: lane-contiguous) ----
  vfloat4 v[kVec];
#pragma unroll
  for (int u = 0; u < kVec; ++u) {
    const uint32_t q = u * kScanThreads + tid;
    if (q < n_vec) v[u] = __builtin_nontemporal_load(src + q);     // streamed once: keep it out of L2's way
    else v[u] = vfloat4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
  }

  // ---- survivor mask: bit (4u+e) <-> element 4*(u*256+tid)+e of the tile (NaN fails >=) ----
  uint64_t mask = 0;
#pragma unroll
  for (int u = 0; u < kVec; ++u) {
    uint32_t m = (v[u].x >= thr ? 1u : 0u) | (v[u].y >= thr ? 2u : 0u) | (v[u].z >= thr ? 4u : 0u) |
                 (v[u].w >= thr ? 8u : 0u);
    mask |= static_cast<uint64_t>(m) << (4 * u);
  }
  const uint32_t cnt = __popcll(mask);

  // position of the tile inside the level: image index and offset within the image
  const uint32_t b0 = static_cast<uint32_t>(tile_base / n);
  const uint32_t r0 = static_cast<uint32_t>(tile_base - static_cast<uint64_t>(b0) * n);
  const bool one_image = static_cast<uint64_t>(r0) + tile_len <= n;

  if (__syncthreads_or(cnt != 0)) {
    if (one_image) {
      // block-exclusive scan of cnt -> one atomic per tile
      const uint32_t inc = wave_inclusive_sum(cnt);
      const int w = tid >> 6;
      if (lane_id() == kWave - 1) s_wave_tot[w] = inc;
      __syncthreads();
      uint32_t wave_off = 0, block_tot = 0;
#pragma unroll
      for (int i = 0; i < kScanThreads / kWave; ++i) {
        const uint32_t t = s_wave_tot[i];
        if (i < w) wave_off += t;
        block_tot += t;
      }
      if (tid == 0) s_base = atomicAdd(a.counts + L.seg_base + b0, block_tot);
      __syncthreads();
      uint32_t slot = s_base + wave_off + inc - cnt;
      uint64_t *dst = a.cand + L.cand_off + static_cast<uint64_t>(b0) * L.cap;
#pragma unroll
      for (int u = 0; u < kVec; ++u) {
        const uint32_t m = static_cast<uint32_t>(mask >> (4 * u)) & 15u;
        if (m) {
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
          const uint32_t idx0 = r0 + 4u * (u * kScanThreads + tid);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (m & (1u << k)) {
              if (slot < L.cap) dst[slot] = make_key(e[k], idx0 + k);
              ++slot;
            }
          }
        }
      }
    } else {
      // tile straddles an image boundary (at most once per image per level): per-survivor atomics
#pragma unroll
      for (int u = 0; u < kVec; ++u) {
        const uint32_t m = static_cast<uint32_t>(mask >> (4 * u)) & 15u;
        if (m) {
          const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (m & (1u << k)) {
              const uint32_t r = r0 + 4u * (u * kScanThreads + tid) + k;
              const uint32_t b = b0 + r / n;
              const uint32_t idx = r % n;
              const uint32_t slot = atomicAdd(a.counts + L.seg_base + b, 1u);
              if (slot < L.cap) a.cand[L.cand_off + static_cast<uint64_t>(b) * L.cap + slot] = make_key(e[k], idx);
            }
          }
        }
      }
    }
  }

  // ---- scalar tail of the level (total % 4 elements, last tile only) ----
  const uint32_t tail = tile_len & 3u;
  if (tail && tid < tail) {
    const uint32_t off = (n_vec << 2) + tid;
    const float s = static_cast<const float *>(L.cls)[tile_base + off];
    if (s >= thr) {
      const uint64_t r = static_cast<uint64_t>(r0) + off;
      const uint32_t b = b0 + static_cast<uint32_t>(r / n);
      const uint32_t idx = static_cast<uint32_t>(r % n);
      const uint32_t slot = atomicAdd(a.counts + L.seg_base + b, 1u);
      if (slot < L.cap) a.cand[L.cand_off + static_cast<uint64_t>(b) * L.cap + slot] = make_key(s, idx);
    }
  }
}

}  // namespace odtk

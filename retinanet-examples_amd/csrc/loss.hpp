// loss.hpp -- fused, masked FocalLoss + SmoothL1 reduction for one pyramid level of the whole batch
// (SURVEY.md 8f rank 3): the step right after target assignment in the training path.
//
// Replaces, per level, the reference's odtk/model.py:193-209 + odtk/loss.py:13-31:
//     cls_loss = FocalLoss(cls_head.view_as(cls_target).float(), cls_target)          (elementwise, ~10 torch kernels)
//     cls_sum  = (cls_loss * (depth >= 0).expand_as(cls_target).float()).sum()
//     box_sum  = (SmoothL1(box_head.float(), box_target) * (depth > 0).float()).sum()
//     fg       = (depth > 0).sum()
// which materialises ~6 elementwise temporaries the size of the classification head (15.4 M logits per
// 800x1280 image) in the forward pass and as many again in autograd's backward.  Here
//   forward  : ONE streaming pass over the logits as the convolution wrote them (fp32 / bf16 / fp16, NCHW or
//              channels_last) -> three scalars.  The one-hot class target is never read: under the mask
//              `depth >= 0` it is exactly `c == depth - 1` (box.py:173-184: background rows are all zero,
//              foreground rows are one-hot at the class, ignored anchors are masked out), so `depth`
//              ([B, A, 1, H, W], 1/C of the head) is all the kernel needs.
//   backward : ONE pass: re-reads the logits, writes d(logits) and d(deltas) in the heads' own dtype and layout,
//              already multiplied by the upstream gradients (read from DEVICE scalars: no host sync).
//
// Roofline: HBM-bound.  Algorithmic bytes per level: forward sizeof(T) per logit; backward 2 x sizeof(T) per
// logit (read + gradient write); depth / box terms are < 2 % of that.
// Arithmetic per element follows loss.py operation by operation in fp32 (the sums are accumulated in fp32 per
// lane over <= 64 elements, then in fp64): forward values agree with the torch expression to ~1e-7 relative.
#pragma once

#include "common.hpp"
#include "prefilter.hpp"   // element types F32 / BF16 / F16, vuint4, load_raw, round_to_*
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kLossThreads = 256;

struct LossArgs {
  const void *cls;          // [B, A*C, H, W] logits, element type T
  const void *box;          // [B, A*NB, H, W] predicted deltas, element type T
  const float *depth;       // [B, A, 1, H, W]  -1 ignore / 0 background / class + 1
  const float *box_target;  // [B, A, NB, H, W]
  double *acc;              // forward: [3] = cls_sum, box_sum, foreground count (atomically accumulated; pre-zeroed)
  const float *g_cls;       // backward: device scalar d(out)/d(cls_sum)   (null: 0)
  const float *g_box;       // backward: device scalar d(out)/d(box_sum)   (null: 0)
  void *dcls;               // backward: gradient w.r.t. cls, same dtype / layout as cls
  void *dbox;               // backward: gradient w.r.t. box, same dtype / layout as box
  uint32_t batch, num_anchors, num_classes, hw, nb;
  uint32_t channels_last;   // layout of cls and box (0: NCHW, 1: NHWC)
  uint32_t cls_blocks;      // blocks [0, cls_blocks) walk the logits, the rest walk the deltas
  float alpha, gamma, beta;
};

// loss.py:13-19 for one element; t is 0 or 1.  Returns the loss, *grad = d(loss)/dx.
// The kernel was ALU-bound on ocml's expf / log1pf / IEEE division (~320 instructions per logit: 990 GB/s); this form
// needs one hardware exp2, one log2 and one reciprocal per logit (each accurate to ~1 ulp, i.e. ~1e-7 relative on the
// terms that carry the sums -- the parity bars are 1e-6 on the sums and 1e-5 on the gradients, tests/test_gpu_loss.py):
//   e = exp(-|x|) in (0, 1];  sigmoid(x) = 1 / (1 + e) for x >= 0, e / (1 + e) for x < 0;  log1p(e) = log(1 + e), 1 + e in (1, 2]
template <bool kGrad>
__device__ __forceinline__ float focal_element(float x, bool positive, float alpha, float gamma, float *grad) {
  const float t = positive ? 1.0f : 0.0f;
  const float e = __expf(-fabsf(x));
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  const float p = x >= 0.0f ? r : e * r;                                        // pred_logits.sigmoid()
  // F.binary_cross_entropy_with_logits: (1 - t) * x + max(-x, 0) + log1p(exp(-|x|))
  const float ce = (1.0f - t) * x + fmaxf(-x, 0.0f) + __logf(1.0f + e);
  const float a_t = t * alpha + (1.0f - t) * (1.0f - alpha);
  const float pt = positive ? p : 1.0f - p;
  const float q = 1.0f - pt;                                                    // (1. - pt)
  const float mod = gamma == 2.0f ? q * q : powf(q, gamma);
  if constexpr (kGrad) {
    // d q / d x = -p (1 - p) for t = 1, +p (1 - p) for t = 0;  d ce / d x = p - t
    const float dq = positive ? -(p * (1.0f - p)) : p * (1.0f - p);
    const float dmod = gamma == 2.0f ? 2.0f * q : (q > 0.0f ? gamma * powf(q, gamma - 1.0f) : 0.0f);
    *grad = a_t * (dmod * dq * ce + mod * (p - t));
  }
  return a_t * mod * ce;
}

// loss.py:27-31
template <bool kGrad>
__device__ __forceinline__ float smooth_l1_element(float pred, float target, float beta, float *grad) {
  const float d = pred - target;
  const float x = fabsf(d);
  if constexpr (kGrad) *grad = x >= beta ? (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) : d / beta;
  return x >= beta ? x - 0.5f * beta : 0.5f * x * x / beta;
}

template <typename T>
__device__ __forceinline__ void store_elem(void *base, uint64_t idx, float v) {
  if constexpr (std::is_same_v<T, F32>) {
    static_cast<float *>(base)[idx] = v;
  } else if constexpr (std::is_same_v<T, BF16>) {
    static_cast<uint16_t *>(base)[idx] = static_cast<uint16_t>(__float_as_uint(round_to_bf16(v)) >> 16);
  } else {
    static_cast<uint16_t *>(base)[idx] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
  }
}

__device__ __forceinline__ double block_sum(double v, double *s_red) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if (lane_id() == 0) s_red[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kLossThreads / kWave; ++i) r += s_red[i];
  return r;                                                                    // valid on thread 0
}

// One workgroup's share of one level: workgroup `block` of the `n_blocks` that level's launch (or its slice of a
// multi-level launch) consists of.  kBackward = false: accumulate the three sums.  kBackward = true: write the gradients.
template <typename T, bool kBackward>
__device__ __forceinline__ void retina_loss_block(const LossArgs &a, uint32_t block, uint32_t n_blocks, double *s_red) {
  constexpr int kPer = T::kPerLoad;
  const uint32_t A = a.num_anchors, C = a.num_classes, hw = a.hw, NB = a.nb;
  const uint32_t channels = A * C;
  float sum_cls = 0.0f, sum_box = 0.0f, n_fg = 0.0f;
  double acc_cls = 0.0, acc_box = 0.0, acc_fg = 0.0;

  if (block < a.cls_blocks) {
    // ---- the logits, in memory order, 16 bytes per lane per trip ----
    const float g = kBackward ? (a.g_cls ? *a.g_cls : 0.0f) : 0.0f;
    // (the host guarantees batch * channels * hw < 2^32: index arithmetic stays in 32 bits -- a 64-bit division
    // costs more than the loss of the whole vector)
    const uint32_t n = a.batch * channels * hw;
    const uint32_t n_vec = n / kPer;
    const vuint4 *src = static_cast<const vuint4 *>(a.cls);
    const uint32_t stride = a.cls_blocks * kLossThreads;
    int trips = 0;
    for (uint32_t v = block * kLossThreads + threadIdx.x; v < n_vec; v += stride) {
      const vuint4 raw = __builtin_nontemporal_load(src + v);
      const uint32_t r0 = v * kPer;
      // decompose the first element once; the others follow by increment with carry
      uint32_t img, an, c, pix;
      if (a.channels_last) {
        const uint32_t p = r0 / channels;
        const uint32_t ch = r0 - p * channels;
        img = p / hw;
        pix = p - img * hw;
        an = ch / C;
        c = ch - an * C;
      } else {
        const uint32_t q = r0 / hw;                          // (img * A + an) * C + c
        pix = r0 - q * hw;
        const uint32_t ia = q / C;
        c = q - ia * C;
        img = ia / A;
        an = ia - img * A;
      }
      // depth of every element of the vector, fetched BEFORE the arithmetic (independent loads, one wait) -- an element's
      // depth read inside the loop put an L2 round trip on every element's critical path.
      //   channels_last, vector inside one anchor's class run (C % kPer == 0: always): ONE depth value
      //   NCHW, vector inside one (image, anchor, class) plane row (hw % kPer == 0: P3..P6): kPer consecutive values
      //   otherwise (tiny levels): element by element with carries
      float dep[kPer];
      uint32_t cls_of[kPer];
      const uint32_t cell = (img * A + an) * hw + pix;
      if (a.channels_last && c + kPer <= C) {
        const float d = a.depth[cell];
#pragma unroll
        for (int e = 0; e < kPer; ++e) { dep[e] = d; cls_of[e] = c + e; }
      } else if (!a.channels_last && pix + kPer <= hw) {
#pragma unroll
        for (int e = 0; e < kPer; ++e) { dep[e] = a.depth[cell + e]; cls_of[e] = c; }
      } else {
        uint32_t i2 = img, a2 = an, c2 = c, p2 = pix;
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
          dep[e] = a.depth[(static_cast<uint64_t>(i2) * A + a2) * hw + p2];
          cls_of[e] = c2;
          if (a.channels_last) { if (++c2 == C) { c2 = 0; if (++a2 == A) { a2 = 0; if (++p2 == hw) { p2 = 0; ++i2; } } } }
          else { if (++p2 == hw) { p2 = 0; if (++c2 == C) { c2 = 0; if (++a2 == A) { a2 = 0; ++i2; } } } }
        }
      }
      float out[kPer];
#pragma unroll
      for (int e = 0; e < kPer; ++e) {
        float x;
        if constexpr (std::is_same_v<T, F32>) {
          x = __uint_as_float(raw[e]);
        } else {
          const uint32_t h = (raw[e >> 1] >> (16 * (e & 1))) & 0xffffu;
          x = std::is_same_v<T, BF16> ? bf16_bits_to_float(h) : f16_bits_to_float(h);
        }
        // evaluated for every element (no branch around the arithmetic: lanes diverge on `dep`), masked afterwards
        const bool positive = dep[e] > 0.0f && static_cast<uint32_t>(dep[e] - 1.0f) == cls_of[e];
        float grad = 0.0f;
        const float l = focal_element<kBackward>(x, positive, a.alpha, a.gamma, &grad);
        const bool counted = dep[e] >= 0.0f;                                    // model.py:199 cls_mask
        if constexpr (!kBackward) sum_cls += counted ? l : 0.0f;
        out[e] = counted ? g * grad : 0.0f;
      }
      if constexpr (kBackward) {
        vuint4 w;
        if constexpr (std::is_same_v<T, F32>) {
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(out[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            uint32_t lo, hi;
            if constexpr (std::is_same_v<T, BF16>) {
              lo = __float_as_uint(round_to_bf16(out[2 * e])) >> 16;
              hi = __float_as_uint(round_to_bf16(out[2 * e + 1])) >> 16;
            } else {
              lo = __builtin_bit_cast(uint16_t, static_cast<_Float16>(out[2 * e]));
              hi = __builtin_bit_cast(uint16_t, static_cast<_Float16>(out[2 * e + 1]));
            }
            w[e] = lo | (hi << 16);
          }
        }
        static_cast<vuint4 *>(a.dcls)[v] = w;
      } else if (++trips == 8) {                                               // fp32 partial of <= 64 elements -> fp64
        acc_cls += sum_cls;
        sum_cls = 0.0f;
        trips = 0;
      }
    }
    // scalar tail (n % kPer elements), first block only
    if (block == 0 && threadIdx.x < n - n_vec * kPer) {
      const uint32_t r = n_vec * kPer + threadIdx.x;
      uint32_t img, an, c, pix;
      if (a.channels_last) {
        const uint32_t p = r / channels, ch = r - p * channels;
        img = p / hw; pix = p - img * hw;
        an = ch / C; c = ch - an * C;
      } else {
        const uint32_t q = r / hw, ia = q / C;
        pix = r - q * hw;
        c = q - ia * C; img = ia / A; an = ia - img * A;
      }
      const float x = load_raw<T>(a.cls, r);
      const float dep = a.depth[(static_cast<uint64_t>(img) * A + an) * hw + pix];
      float grad = 0.0f;
      if (dep >= 0.0f) {
        const bool positive = dep > 0.0f && static_cast<uint32_t>(dep - 1.0f) == c;
        const float l = focal_element<kBackward>(x, positive, a.alpha, a.gamma, &grad);
        if constexpr (!kBackward) sum_cls += l;
      }
      if constexpr (kBackward) store_elem<T>(a.dcls, r, g * grad);
    }
    acc_cls += sum_cls;
  } else {
    // ---- the box deltas: one lane per (image, anchor, pixel), NB parameters each; only foreground anchors count ----
    const float g = kBackward ? (a.g_box ? *a.g_box : 0.0f) : 0.0f;
    const uint64_t cells = static_cast<uint64_t>(a.batch) * A * hw;
    const uint64_t stride = static_cast<uint64_t>(n_blocks - a.cls_blocks) * kLossThreads;
    for (uint64_t cell = static_cast<uint64_t>(block - a.cls_blocks) * kLossThreads + threadIdx.x; cell < cells; cell += stride) {
      const uint64_t ia = cell / hw;                          // img * A + an
      const uint32_t pix = static_cast<uint32_t>(cell - ia * hw);
      const uint32_t img = static_cast<uint32_t>(ia / A), an = static_cast<uint32_t>(ia - static_cast<uint64_t>(img) * A);
      const bool fg = a.depth[cell] > 0.0f;                                     // model.py:204 box_mask
      if (!kBackward && fg) n_fg += 1.0f;
      if (!fg && !kBackward) continue;
      for (uint32_t k = 0; k < NB; ++k) {
        const uint64_t off = a.channels_last ? (static_cast<uint64_t>(img) * hw + pix) * (A * NB) + an * NB + k
                                             : (ia * NB + k) * hw + pix;
        float grad = 0.0f;
        if (fg) {
          const float l = smooth_l1_element<kBackward>(load_raw<T>(a.box, off), a.box_target[(ia * NB + k) * hw + pix], a.beta, &grad);
          if constexpr (!kBackward) sum_box += l;
        }
        if constexpr (kBackward) store_elem<T>(a.dbox, off, g * grad);
      }
    }
    acc_box += sum_box;
    acc_fg += n_fg;
  }

  if constexpr (!kBackward) {
    const double c_ = block_sum(acc_cls, s_red);
    const double b_ = block_sum(acc_box, s_red);
    const double f_ = block_sum(acc_fg, s_red);
    if (threadIdx.x == 0) {
      if (c_ != 0.0) atomicAdd(a.acc + 0, c_);
      if (b_ != 0.0) atomicAdd(a.acc + 1, b_);
      if (f_ != 0.0) atomicAdd(a.acc + 2, f_);
    }
  }
}

template <typename T, bool kBackward>
__global__ __launch_bounds__(kLossThreads) void retina_loss_kernel(const LossArgs a) {
  __shared__ double s_red[kLossThreads / kWave];
  retina_loss_block<T, kBackward>(a, blockIdx.x, gridDim.x, s_red);
}

// All pyramid levels of the batch in ONE launch per direction (ten launches per training step become two): the
// workgroups of level l are [block_begin[l], block_begin[l + 1]).
struct LossLevelsArgs {
  LossArgs lv[ODTK_MAX_LEVELS];
  uint32_t block_begin[ODTK_MAX_LEVELS + 1];
  int n_levels;
};

template <typename T, bool kBackward>
__global__ __launch_bounds__(kLossThreads) void retina_loss_levels_kernel(const LossLevelsArgs a) {
  __shared__ double s_red[kLossThreads / kWave];
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.block_begin[i]) l = i;
  retina_loss_block<T, kBackward>(a.lv[l], blockIdx.x - a.block_begin[l], a.block_begin[l + 1] - a.block_begin[l], s_red);
}

}  // namespace odtk

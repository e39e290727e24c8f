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
// Measured (round 3, tools/loss_probe.py, 2 images of 800x1280, cold inputs): fp32 forward 33 us (3.7 TB/s),
// backward 52 us (4.8 TB/s); in the training step 88.7 us for both = 0.52 of 8 TB/s (round 2: 124.7 us, 0.37).
// The forward is VALU-bound (fp16 heads take the same 30 us as fp32 ones, and an atomic-free reduction -- the workspace form
// below -- measures the same): ~27 issue slots of arithmetic + ~5 of index math per logit, three of the instructions
// quarter-rate (exp2, rcp, log2), against 39 T lane-operations/s.  Launch shapes: odtk_debug_loss_tuning, defaults in
// odtk_hip.hip -- the forward with atomics wants few, large workgroups (every workgroup ends in a double atomic on its
// level's word: 2048 workgroups per level cost +80 us of queueing).
// Arithmetic per element: loss.py's expressions in fp32, in the symmetric form derived at focal_term below (the sums
// are accumulated in fp32 per lane over <= 32 elements, then in fp64): forward values agree with the torch expression
// evaluated in float64 to 1e-6 relative, gradients to 1e-5 of the largest gradient (tests/test_gpu_loss.py).
// Index arithmetic: offset -> (image, anchor, class, pixel) by multiply-high (fastdiv.hpp), three per 16-byte vector.
#pragma once

#include <type_traits>

#include "common.hpp"
#include "fastdiv.hpp"
#include "prefilter.hpp"   // element types F32 / BF16 / F16, vuint4, load_raw, round_to_*
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kLossMaxThreads = 1024;         // the kernels take any workgroup size that is a multiple of 64 up to this

struct LossArgs {
  const void *cls;          // [B, A*C, H, W] logits, element type T
  const void *box;          // [B, A*NB, H, W] predicted deltas, element type T
  const float *depth;       // [B, A, 1, H, W]  -1 ignore / 0 background / class + 1
  const float *box_target;  // [B, A, NB, H, W]
  double *acc;              // forward: [3] = cls_sum, box_sum, foreground count (atomically accumulated; pre-zeroed)
  double *partial;          // forward, workspace form: [gridDim.x][3] per-workgroup sums of the whole launch (acc unused);
                            // loss_reduce_kernel adds them up afterwards -- no atomics, one fixed summation order
  const float *g_cls;       // backward: device scalar d(out)/d(cls_sum)   (null: 0)
  const float *g_box;       // backward: device scalar d(out)/d(box_sum)   (null: 0)
  void *dcls;               // backward: gradient w.r.t. cls, same dtype / layout as cls
  void *dbox;               // backward: gradient w.r.t. box, same dtype / layout as box
  uint32_t batch, num_anchors, num_classes, hw, nb;
  uint32_t channels_last;   // layout of cls and box (0: NCHW, 1: NHWC)
  uint32_t cls_blocks;      // blocks [0, cls_blocks) walk the logits, the rest walk the deltas
  uint32_t per_wave;        // forward, workspace form: 1 = every WAVE writes its own three sums (partial is [gridDim.x * waves][3]):
                            // no workgroup barrier, the waves retire on their own as the prefilter's do
  uint32_t box_rows;        // backward, channels_last: 1 = the box-delta walk follows d(deltas)' memory order (one vector store per cell)
  uint32_t window;          // logit walk: 0 = a trip's kUnroll vectors lie `cls_blocks x blockDim` vectors apart (one window per
                            // vector, every workgroup in every window); 1 = a wave's kUnroll vectors of a trip are contiguous
  float alpha, gamma, beta;
  FastDiv by_channels, by_hw, by_classes, by_anchors;   // A*C, H*W, C, A (host: fastdiv_make)
};

// loss.py:13-19 for one element with target t in {0, 1}.  kBackward = false: the loss times `w`; true: d(loss)/dx times `w`
// (the caller folds alpha_t -- and, backward, the upstream gradient and the sign below -- into w_neg / w_pos).
//
// Round 3.  The loss is symmetric under (x, t) -> (-x, 1 - t): with s = x for t = 0 and s = -x for t = 1,
//     q  = sigmoid(s)            = 1 - pt                       (loss.py:16-17)
//     ce = softplus(s)           = BCE-with-logits(x, t)        (loss.py:15: (1-t) x + max(-x, 0) + log1p(exp(-|x|)))
//     loss = alpha_t q^gamma ce,  d loss / d s = alpha_t q^gamma (gamma (1 - q) ce + q),  d s / d x = +1 (t = 0), -1 (t = 1)
// so one form serves both targets, `1 - q` comes out of the same reciprocal without a cancelling subtraction (more
// accurate than the reference's fp32 `1 - pt` where pt -> 1), and an element costs one hardware exp2, one rcp, one log2
// (each ~1 ulp; the parity bars are 1e-6 on the sums and 1e-5 on the gradients, tests/test_gpu_loss.py) plus ~15
// full-rate VALU operations: ~27 issue slots forward, ~31 backward.  (History: ocml expf / log1pf / IEEE division, ~320
// instructions per logit, 990 GB/s -> hardware transcendentals in the reference's own operation order, 2.9 TB/s.)
template <bool kBackward, bool kGamma2>
__device__ __forceinline__ float focal_term(float x, bool positive, float w_neg, float w_pos, float gamma) {
  const float s = positive ? -x : x;
  const float e = __builtin_amdgcn_exp2f(fabsf(s) * -1.4426950408889634f);      // exp(-|s|) in (0, 1]
  const float d = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(d);
  const float er = e * r;
  const bool nonneg = s >= 0.0f;
  const float q = nonneg ? r : er;                                               // sigmoid(s)
  const float ce = fmaf(__builtin_amdgcn_logf(d), 0.6931471805599453f, fmaxf(s, 0.0f));   // softplus(s); 1 + e in (1, 2]
  const float w = positive ? w_pos : w_neg;
  // q^gamma: q in [0, 1]; the general form is exp2(gamma log2 q) on the hardware units.  log2 0 = -inf is held at -150 so
  // that gamma = 0 gives 0^0 = 1 as torch's pow does (0 * -inf would be NaN); 2^(-150 gamma) is 0 or, for small gamma, a
  // factor that multiplies a cross entropy of exactly 0 (q == 0 means exp(-|s|) underflowed: s < -87)
  const float mod = kGamma2 ? q * q : __builtin_amdgcn_exp2f(gamma * fmaxf(__builtin_amdgcn_logf(q), -150.0f));
  if constexpr (!kBackward) {
    return w * mod * ce;
  } else {
    const float omq = nonneg ? er : r;                                           // 1 - sigmoid(s), no cancellation
    return w * mod * fmaf((kGamma2 ? 2.0f : gamma) * omq, ce, q);
  }
}

// Round 4 (late): the form for a vector of NEGATIVES.  All but about one 16-byte vector in a thousand hold no element that is
// its anchor's class (foreground anchors are ~0.5 % of the cells and one class of C each) and no logit beyond kPlainMax; for
// those neither the target select nor the split on the sign of s is needed.  With u = exp(x):
//     q = sigmoid(x) = u / (1 + u),   1 - q = 1 / (1 + u)  (no cancellation),   ce = softplus(x) = ln(1 + u)
//     loss = (1 - alpha) q^2 ce,      d loss / dx = (1 - alpha) q^2 (2 (1 - q) ce + q)          (focal_term with t = 0, gamma = 2)
// -- the same three hardware transcendentals, 6 (forward) / 9 (backward) full-rate operations instead of ~15 / ~19; the
// weight (1 - alpha) (x ln 2 forward, x the upstream gradient backward) is applied by the caller.  u stays finite for
// x <= kPlainMax (exp(64) = 6e27) and underflows to 0 for x < -87, where q = 0 and the term is 0 as in focal_term.
// Vectors with a positive element or a logit beyond kPlainMax (+inf included) take focal_term element by element as before.
// A NaN logit is ignored by the maximum and yields NaN in either form; -inf gives u = 0: term and gradient 0, as focal_term.
constexpr float kPlainMax = 64.0f;

template <bool kBackward>
__device__ __forceinline__ float focal_plain(float x) {
  const float u = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  const float d = 1.0f + u;
  const float r = __builtin_amdgcn_rcpf(d);                                      // 1 - sigmoid(x)
  const float q = u * r;                                                         // sigmoid(x)
  const float l2 = __builtin_amdgcn_logf(d);                                     // softplus(x) / ln 2
  if constexpr (!kBackward) return q * q * l2;                                   // x (1 - alpha) ln 2
  else return q * q * fmaf(r * l2, 1.3862943611198906f, q);                      // x (1 - alpha) g      (2 ln 2)
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

// four consecutive elements, element index 4 * idx4 (the base is 16-byte aligned: checked by the host)
template <typename T>
__device__ __forceinline__ void store_vec4(void *base, uint64_t idx4, const float *v) {
  if constexpr (std::is_same_v<T, F32>) {
    static_cast<vuint4 *>(base)[idx4] = vuint4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  } else {
    uint32_t h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (std::is_same_v<T, BF16>) h[e] = __float_as_uint(round_to_bf16(v[e])) >> 16;
      else h[e] = __builtin_bit_cast(uint16_t, static_cast<_Float16>(v[e]));
    }
    static_cast<uint2 *>(base)[idx4] = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
  }
}

template <typename T>
__device__ __forceinline__ float vec_elem(const vuint4 &raw, int e) {
  if constexpr (std::is_same_v<T, F32>) {
    return __uint_as_float(raw[e]);
  } else {
    const uint32_t w = raw[e >> 1];
    if constexpr (std::is_same_v<T, BF16>) return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
    else return f16_bits_to_float((e & 1) ? (w >> 16) : (w & 0xffffu));
  }
}

template <typename T>
__device__ __forceinline__ vuint4 pack_vec(const float *out) {
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
  return w;
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
    for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) r += s_red[i];
  return r;                                                                    // valid on thread 0
}

// The logits of one level, in memory order, 16 bytes per lane per vector, kUnroll vectors per trip: all of a trip's
// loads (logits AND the depth words they need) are issued before any arithmetic -- the kernel is a stream with ~30 issue
// slots of work per element, and at one 16-byte load per wave in flight (round 2) it was latency-bound at 0.37 of HBM.
//   kCL (channels_last), vector inside one anchor's class run (C % kPer == 0: always): ONE depth value per vector
//   NCHW, vector inside one (image, anchor, class) plane row (hw % kPer == 0: P3..P6): kPer consecutive depth values
//   otherwise (tiny levels, class counts that are no multiple of the vector): element by element with carries
//   kForm (gamma = 2 only; the launch-time switch is odtk_debug_loss_form): 0 = every element through focal_term, 1 = fast
//   vectors of negatives take focal_plain (above).  2..4 are TIMING ABLATIONS of form 1 whose results are wrong on purpose
//   (tools/loss_form_probe.py; fp32 forward only): 2 = no depth gather (every cell counts as background), 3 = no arithmetic
//   (the logits are added up as they are), 4 = no index arithmetic and no depth gather, 6 = the box-delta workgroups return at
//   once, 7 = the logit workgroups return at once.
template <typename T, bool kBackward, bool kGamma2, bool kCL, int kUnroll, int kForm>
__device__ __forceinline__ double focal_stream(const LossArgs &a, uint32_t block) {
  static_assert(kForm == 0 || kGamma2, "focal_plain is the gamma = 2 form");
  constexpr bool kPlain = kForm >= 1, kNoDepth = kForm == 2 || kForm == 4, kNoIndex = kForm == 4, kNoMath = kForm == 3;
  constexpr int kPer = T::kPerLoad;
  constexpr int kDep = kCL ? 1 : kPer;                     // depth words per fast vector
  const uint32_t A = a.num_anchors, C = a.num_classes, hw = a.hw;
  const uint32_t channels = A * C;
  const float g = kBackward ? (a.g_cls ? *a.g_cls : 0.0f) : 1.0f;
  const float w_neg = (1.0f - a.alpha) * g, w_pos = kBackward ? -(a.alpha * g) : a.alpha;   // loss.py:18 alpha_t (x ds/dx)
  const float gamma = a.gamma;
  // (the host guarantees batch * channels * hw < 2^32: index arithmetic stays in 32 bits)
  const uint32_t n = a.batch * channels * hw;
  const uint32_t n_vec = n / kPer;
  const vuint4 *src = static_cast<const vuint4 *>(a.cls);
  const uint32_t stride = a.cls_blocks * blockDim.x;
  const uint32_t ustride = a.window ? static_cast<uint32_t>(kWave) : stride;   // between the vectors of a trip (launch-uniform)
  const uint32_t first = a.window ? (block * blockDim.x + (threadIdx.x & ~static_cast<uint32_t>(kWave - 1))) * kUnroll + (threadIdx.x & (kWave - 1))
                                  : block * blockDim.x + threadIdx.x;
  double acc = 0.0;

  for (uint32_t v0 = first; v0 < n_vec; v0 += stride * kUnroll) {
    vuint4 raw[kUnroll];
    float dep[kUnroll][kDep];
    uint32_t c0[kUnroll];                                  // class of the vector's first element
    bool fast[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t v = v0 + u * ustride;
      fast[u] = false;
      c0[u] = 0;
#pragma unroll
      for (int e = 0; e < kDep; ++e) dep[u][e] = -1.0f;
      if (v < n_vec) {
        raw[u] = __builtin_nontemporal_load(src + v);
        const uint32_t r0 = v * kPer;
        if constexpr (kNoIndex) {
          fast[u] = true;
          dep[u][0] = 0.0f;
          if constexpr (!kCL) {
#pragma unroll
            for (int e = 0; e < kDep; ++e) dep[u][e] = 0.0f;
          }
        } else if constexpr (kCL) {
          uint32_t ch, pix, c;
          const uint32_t p = fastdivmod(r0, a.by_channels, &ch);
          const uint32_t img = fastdivmod(p, a.by_hw, &pix);
          const uint32_t an = fastdivmod(ch, a.by_classes, &c);
          c0[u] = c;
          fast[u] = c + kPer <= C;
          if (fast[u]) dep[u][0] = kNoDepth ? ((img * A + an) * hw + pix == 0xffffffffu ? -1.0f : 0.0f) : a.depth[(img * A + an) * hw + pix];
        } else {
          uint32_t pix, c;
          const uint32_t q = fastdivmod(r0, a.by_hw, &pix);           // (img * A + an) * C + c
          const uint32_t ia = fastdivmod(q, a.by_classes, &c);
          c0[u] = c;
          fast[u] = pix + kPer <= hw;
          if (fast[u]) {
#pragma unroll
            for (int e = 0; e < kDep; ++e) dep[u][e] = kNoDepth ? (ia * hw + pix + e == 0xffffffffu ? -1.0f : 0.0f) : a.depth[ia * hw + pix + e];
          }
        }
      } else {
        raw[u] = vuint4{0u, 0u, 0u, 0u};
      }
    }
    float sum = 0.0f;                                      // fp32 partial of <= kUnroll * kPer <= 32 elements -> fp64
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t v = v0 + u * ustride;
      if (v >= n_vec) break;
      float out[kPer];
      if (fast[u]) {
        // evaluated for every element (no branch around the arithmetic: lanes diverge on `depth`), masked afterwards.
        // depth is integral by contract (-1 / 0 / class + 1): "depth > 0 and class == depth - 1" is ONE compare
        const float tgt0 = static_cast<float>(c0[u] + 1);
        float vs = 0.0f;
        bool plain = false;
        if constexpr (kPlain) {
          float mx = vec_elem<T>(raw[u], 0);
#pragma unroll
          for (int e = 1; e < kPer; ++e) mx = fmaxf(mx, vec_elem<T>(raw[u], e));
          if constexpr (kCL) {
            const float rel = dep[u][0] - tgt0;                                  // which element is the anchor's class, if any
            plain = !(rel >= 0.0f && rel < static_cast<float>(kPer));
          } else {
            plain = true;
#pragma unroll
            for (int e = 0; e < kPer; ++e) plain = plain && dep[u][e] != tgt0;
          }
          plain = plain && mx <= kPlainMax;
        }
        if (plain) {
          // a vector of negatives (or ignored cells): one weight for the whole vector
          const float wv = kBackward ? w_neg : w_neg * 0.6931471805599453f;
#pragma unroll
          for (int e = 0; e < kPer; ++e) {
            const float t = kNoMath ? vec_elem<T>(raw[u], e) : focal_plain<kBackward>(vec_elem<T>(raw[u], e));
            if constexpr (kBackward) out[e] = dep[u][kCL ? 0 : e] >= 0.0f ? wv * t : 0.0f;
            else if constexpr (kCL) vs += t;
            else vs += dep[u][e] >= 0.0f ? t : 0.0f;
          }
          if constexpr (!kBackward) sum += kCL ? (dep[u][0] >= 0.0f ? wv * vs : 0.0f) : wv * vs;
        } else {
#pragma unroll
          for (int e = 0; e < kPer; ++e) {
            const float d = dep[u][kCL ? 0 : e];
            const bool positive = d == (kCL ? tgt0 + static_cast<float>(e) : tgt0);
            const float t = focal_term<kBackward, kGamma2>(vec_elem<T>(raw[u], e), positive, w_neg, w_pos, gamma);
            if constexpr (kBackward) out[e] = d >= 0.0f ? t : 0.0f;              // model.py:199 cls_mask
            else if constexpr (kCL) vs += t;
            else vs += d >= 0.0f ? t : 0.0f;
          }
          if constexpr (!kBackward) sum += kCL ? (dep[u][0] >= 0.0f ? vs : 0.0f) : vs;
        }
      } else {
        // decompose the first element again; the others follow by increment with carry
        const uint32_t r0 = v * kPer;
        uint32_t i2, a2, c2, p2;
        if constexpr (kCL) {
          uint32_t ch;
          const uint32_t p = fastdivmod(r0, a.by_channels, &ch);
          i2 = fastdivmod(p, a.by_hw, &p2);
          a2 = fastdivmod(ch, a.by_classes, &c2);
        } else {
          const uint32_t q = fastdivmod(r0, a.by_hw, &p2);
          const uint32_t ia = fastdivmod(q, a.by_classes, &c2);
          i2 = fastdivmod(ia, a.by_anchors, &a2);
        }
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
          const float d = a.depth[(i2 * A + a2) * hw + p2];
          const bool positive = d == static_cast<float>(c2 + 1);
          const float t = focal_term<kBackward, kGamma2>(vec_elem<T>(raw[u], e), positive, w_neg, w_pos, gamma);
          if constexpr (kBackward) out[e] = d >= 0.0f ? t : 0.0f;
          else sum += d >= 0.0f ? t : 0.0f;
          if constexpr (kCL) { if (++c2 == C) { c2 = 0; if (++a2 == A) { a2 = 0; if (++p2 == hw) { p2 = 0; ++i2; } } } }
          else { if (++p2 == hw) { p2 = 0; if (++c2 == C) { c2 = 0; if (++a2 == A) { a2 = 0; ++i2; } } } }
        }
      }
      if constexpr (kBackward) __builtin_nontemporal_store(pack_vec<T>(out), static_cast<vuint4 *>(a.dcls) + v);
    }
    if constexpr (!kBackward) acc += sum;
  }

  // scalar tail (n % kPer elements), first block only
  if (block == 0 && threadIdx.x < n - n_vec * kPer) {
    const uint32_t r = n_vec * kPer + threadIdx.x;
    uint32_t img, an, c, pix;
    if constexpr (kCL) {
      uint32_t ch;
      const uint32_t p = fastdivmod(r, a.by_channels, &ch);
      img = fastdivmod(p, a.by_hw, &pix);
      an = fastdivmod(ch, a.by_classes, &c);
    } else {
      const uint32_t q = fastdivmod(r, a.by_hw, &pix);
      const uint32_t ia = fastdivmod(q, a.by_classes, &c);
      img = fastdivmod(ia, a.by_anchors, &an);
    }
    const float d = a.depth[(img * A + an) * hw + pix];
    const float t = focal_term<kBackward, kGamma2>(load_raw<T>(a.cls, r), d == static_cast<float>(c + 1), w_neg, w_pos, gamma);
    if constexpr (kBackward) store_elem<T>(a.dcls, r, d >= 0.0f ? t : 0.0f);
    else acc += d >= 0.0f ? t : 0.0f;
  }
  return acc;
}

// One workgroup's share of one level: workgroup `block` of the `n_blocks` that level's slice of the launch consists of.
// kBackward = false: accumulate the three sums.  kBackward = true: write the gradients.
template <typename T, bool kBackward, int kUnroll, int kForm>
__device__ __forceinline__ void retina_loss_block(const LossArgs &a, uint32_t block, uint32_t n_blocks, double *s_red) {
  const uint32_t A = a.num_anchors, hw = a.hw, NB = a.nb;
  double acc_cls = 0.0, acc_box = 0.0, acc_fg = 0.0;

  if (block < a.cls_blocks) {
    if constexpr (kForm == 7) return;                        // timing ablation: no logit walk (wrong sums on purpose)
    const bool g2 = a.gamma == 2.0f;                        // launch-uniform
    if (a.channels_last) acc_cls = g2 ? focal_stream<T, kBackward, true, true, kUnroll, kForm>(a, block)
                                      : focal_stream<T, kBackward, false, true, kUnroll, 0>(a, block);
    else acc_cls = g2 ? focal_stream<T, kBackward, true, false, kUnroll, kForm>(a, block)
                      : focal_stream<T, kBackward, false, false, kUnroll, 0>(a, block);
  } else {
    // ---- the box deltas: one lane per (image, anchor, pixel), NB parameters each; only foreground anchors count ----
    if constexpr (kForm == 6) return;                        // timing ablation: no box-delta walk (wrong sums on purpose)
    const float g = kBackward ? (a.g_box ? *a.g_box : 0.0f) : 0.0f;
    float sum_box = 0.0f, n_fg = 0.0f;
    const uint32_t cells = a.batch * A * hw;                // < 2^32 (host)
    const uint32_t stride = (n_blocks - a.cls_blocks) * blockDim.x;
    if (kBackward && a.channels_last && a.box_rows) {
      // channels_last gradients (round 6): lanes enumerate the cells in the order d(deltas) lies in memory -- (image, pixel,
      // anchor), NB consecutive parameters each -- so a wave's stores are one contiguous run (NB = 4: ONE 16- / 8-byte
      // vector per lane) instead of NB stores of one element each, 4 A NB bytes apart from lane to lane: those were 64
      // partial-line writes per wave and store, as many L2 requests again as half the logit stream
      // (profiles/r05_loss_pmc.txt: 4.56 M requests in the backward for 2.9 M of logits).  depth / box_target are gathered
      // (planes of hw floats); only foreground cells (~0.5 %) read the deltas and the targets at all.
      for (uint32_t cell = (block - a.cls_blocks) * blockDim.x + threadIdx.x; cell < cells; cell += stride) {
        uint32_t pix, an;
        const uint32_t ip = fastdivmod(cell, a.by_anchors, &an);   // img * hw + pix
        const uint32_t img = fastdivmod(ip, a.by_hw, &pix);
        const uint32_t ia = img * A + an;
        const bool fg = a.depth[ia * hw + pix] > 0.0f;                          // model.py:204 box_mask
        const uint64_t off = static_cast<uint64_t>(cell) * NB;
        if (NB == 4) {                                                           // (launch-uniform)
          float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          if (fg) {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
              float grad;
              smooth_l1_element<true>(load_raw<T>(a.box, off + k), a.box_target[(static_cast<uint64_t>(ia) * 4 + k) * hw + pix], a.beta, &grad);
              out[k] = g * grad;
            }
          }
          store_vec4<T>(a.dbox, cell, out);
        } else {
          for (uint32_t k = 0; k < NB; ++k) {
            float grad = 0.0f;
            if (fg) smooth_l1_element<true>(load_raw<T>(a.box, off + k), a.box_target[(static_cast<uint64_t>(ia) * NB + k) * hw + pix], a.beta, &grad);
            store_elem<T>(a.dbox, off + k, g * grad);
          }
        }
      }
    } else {
    for (uint32_t cell = (block - a.cls_blocks) * blockDim.x + threadIdx.x; cell < cells; cell += stride) {
      uint32_t pix, an;
      const uint32_t ia = fastdivmod(cell, a.by_hw, &pix);  // img * A + an
      const uint32_t img = fastdivmod(ia, a.by_anchors, &an);
      const bool fg = a.depth[cell] > 0.0f;                                     // model.py:204 box_mask
      if (!kBackward && fg) n_fg += 1.0f;
      if (!fg && !kBackward) continue;
      for (uint32_t k = 0; k < NB; ++k) {
        const uint64_t off = a.channels_last ? (static_cast<uint64_t>(img) * hw + pix) * (A * NB) + an * NB + k
                                             : (static_cast<uint64_t>(ia) * NB + k) * hw + pix;
        float grad = 0.0f;
        if (fg) {
          const float l = smooth_l1_element<kBackward>(load_raw<T>(a.box, off),
                                                       a.box_target[(static_cast<uint64_t>(ia) * NB + k) * hw + pix], a.beta, &grad);
          if constexpr (!kBackward) sum_box += l;
        }
        if constexpr (kBackward) store_elem<T>(a.dbox, off, g * grad);
      }
    }
    }
    acc_box += sum_box;
    acc_fg += n_fg;
  }

  if constexpr (!kBackward) {
    if (a.partial && a.per_wave) {                           // (launch-uniform) no barrier: wave sums straight to the workspace
      double v3[3] = {acc_cls, acc_box, acc_fg};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v3[k] += __shfl_xor(v3[k], d, kWave);
      }
      if (lane_id() == 0) {
        double *mine = a.partial + 3 * (static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6));
        mine[0] = v3[0]; mine[1] = v3[1]; mine[2] = v3[2];
      }
      return;
    }
    const double c_ = block_sum(acc_cls, s_red);
    const double b_ = block_sum(acc_box, s_red);
    const double f_ = block_sum(acc_fg, s_red);
    if (threadIdx.x == 0) {
      if (a.partial) {
        double *mine = a.partial + 3 * static_cast<size_t>(blockIdx.x);
        mine[0] = c_; mine[1] = b_; mine[2] = f_;
      } else {
        if (c_ != 0.0) atomicAdd(a.acc + 0, c_);
        if (b_ != 0.0) atomicAdd(a.acc + 1, b_);
        if (f_ != 0.0) atomicAdd(a.acc + 2, f_);
      }
    }
  }
}

// One or all pyramid levels of the batch in ONE launch per direction (ten launches per training step become two): the
// workgroups of level l are [block_begin[l], block_begin[l + 1]).
struct LossLevelsArgs {
  LossArgs lv[ODTK_MAX_LEVELS];
  uint32_t block_begin[ODTK_MAX_LEVELS + 1];
  int n_levels;
};

template <typename T, bool kBackward, int kUnroll, int kForm>
__global__ __launch_bounds__(kLossMaxThreads) void retina_loss_kernel(const LossLevelsArgs a) {
  __shared__ double s_red[kLossMaxThreads / kWave];
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.block_begin[i]) l = i;
  retina_loss_block<T, kBackward, kUnroll, kForm>(a.lv[l], blockIdx.x - a.block_begin[l], a.block_begin[l + 1] - a.block_begin[l], s_red);
}

// Second launch of the workspace form of the forward: workgroup l adds up the per-workgroup sums of level l in a fixed
// order (thread t takes partials t, t + 1024, ...; then the wave / block tree) -> sums[l][0..2].  The loss is then bitwise
// reproducible from run to run, which the atomic form is not.
struct LossReduceArgs {
  const double *partial;                       // [total workgroups x per][3]
  uint32_t per;                                // partial sums per workgroup: 1, or its waves (LossArgs.per_wave)
  double *sums;                                // [n_levels][3]
  uint32_t block_begin[ODTK_MAX_LEVELS + 1];
};

constexpr int kLossReduceThreads = 1024;

__global__ __launch_bounds__(kLossReduceThreads) void loss_reduce_kernel(const LossReduceArgs a) {
  __shared__ double s_red[kLossReduceThreads / kWave];
  const uint32_t l = blockIdx.x, b0 = a.block_begin[l] * a.per, b1 = a.block_begin[l + 1] * a.per;
  double v[3] = {0.0, 0.0, 0.0};
  // four partial triples per thread and trip, all twelve loads in flight before the first addition (one workgroup per level
  // reads up to tens of thousands of triples: with one dependent round trip per triple the launch took longer than the walk)
  constexpr uint32_t kT = kLossReduceThreads;
  for (uint32_t b = b0 + threadIdx.x; b < b1; b += 4 * kT) {
    double t[4][3];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      const uint32_t i = b + u * kT;
#pragma unroll
      for (int k = 0; k < 3; ++k) t[u][k] = i < b1 ? a.partial[3 * static_cast<size_t>(i) + k] : 0.0;
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] += t[u][k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double r = block_sum(v[k], s_red);
    if (threadIdx.x == 0) a.sums[3 * l + k] = r;
  }
}

}  // namespace odtk

// epilogue.hpp -- fused per-channel bias (+ residual) (+ ReLU) over a channels_last activation, in
// place: the inference-time epilogue of every convolution of the ResNet/FPN/head stack.
//
// No reference equivalent as a kernel: in the reference (odtk/backbones/resnet.py via torchvision
// blocks, odtk/backbones/layers.py:5-16 FixedBatchNorm2d, odtk/model.py:57-62 heads) each of
// conv-bias, batch-norm, residual add and ReLU is its own full read+write pass over the activation
// (measured on MI355X: those passes cost MORE than the MIOpen convolutions between them:
// 1x1 64->256 @200x320 bs8: conv 71 us, +bias+relu 280 us).  With the frozen BN folded into the
// convolution weights (scale) and this epilogue (shift), a conv->BN->(+skip)->ReLU group reads the
// conv output once (+ the skip once) and writes once.
//
// HBM-bound: algorithmic bytes = 2 x sizeof(T) per element (+ sizeof(T) with a residual).
// 16-byte vector accesses, kUnroll independent vectors in flight per lane, fp32 arithmetic,
// round-to-nearest-even to the storage type.  In the fast form the grid stride is a multiple of the
// row length, so each lane's channel offset never changes: its bias values are loaded ONCE into
// registers and the loop has no integer division.
#pragma once

#include "common.hpp"
#include "fastdiv.hpp"
#include "prefilter.hpp"   // element types, bf16/f16 conversions

namespace odtk {

template <typename T>
__device__ __forceinline__ uint32_t float_to_storage(float f) {
  if constexpr (std::is_same_v<T, BF16>) return __float_as_uint(round_to_bf16(f)) >> 16;
  else return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<_Float16>(f)));
}

template <typename T, bool kResidual, bool kRelu>
__device__ __forceinline__ vuint4 bias_act_vec(vuint4 v, vuint4 r, const float *b) {
  constexpr int kPer = T::kPerLoad;
  float f[kPer];
#pragma unroll
  for (int e = 0; e < kPer; ++e) {
    float x, s = 0.0f;
    if constexpr (std::is_same_v<T, F32>) {
      x = __uint_as_float(v[e]);
      if (kResidual) s = __uint_as_float(r[e]);
    } else {
      const uint32_t hx = (v[e >> 1] >> (16 * (e & 1))) & 0xffffu, hr = (r[e >> 1] >> (16 * (e & 1))) & 0xffffu;
      x = std::is_same_v<T, BF16> ? bf16_bits_to_float(hx) : f16_bits_to_float(hx);
      if (kResidual) s = std::is_same_v<T, BF16> ? bf16_bits_to_float(hr) : f16_bits_to_float(hr);
    }
    float o = x + b[e];
    if (kResidual) o += s;
    if (kRelu) o = o > 0.0f ? o : (o != o ? o : 0.0f);     // NaN goes through, as torch.relu (round 6: it became 0 before)
    f[e] = o;
  }
  if constexpr (std::is_same_v<T, F32>) {
#pragma unroll
    for (int e = 0; e < kPer; ++e) v[e] = __float_as_uint(f[e]);
  } else {
#pragma unroll
    for (int e = 0; e < kPer; e += 2) v[e >> 1] = float_to_storage<T>(f[e]) | (float_to_storage<T>(f[e + 1]) << 16);
  }
  return v;
}

// Fast form.  Requires channels % kPer == 0 and (gridDim.x * blockDim.x) % (channels / kPer) == 0.
template <typename T, bool kResidual, bool kRelu>
__global__ __launch_bounds__(256) void bias_act_kernel(void *y, const float *__restrict__ bias, const void *res,
                                                       uint64_t n_vec, uint32_t vec_per_row) {
  constexpr int kPer = T::kPerLoad;
  constexpr int kUnroll = 4;
  vuint4 *yv = static_cast<vuint4 *>(y);
  const vuint4 *rv = static_cast<const vuint4 *>(res);
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  const uint64_t q0 = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  float b[kPer];
  {
    const uint32_t c0 = static_cast<uint32_t>(q0 % vec_per_row) * kPer;       // loop-invariant
#pragma unroll
    for (int e = 0; e < kPer; ++e) b[e] = bias[c0 + e];
  }
  uint64_t q = q0;
  for (; q + (kUnroll - 1) * step < n_vec; q += kUnroll * step) {
    vuint4 v[kUnroll], r[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      v[u] = yv[q + u * step];
      r[u] = kResidual ? rv[q + u * step] : vuint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) yv[q + u * step] = bias_act_vec<T, kResidual, kRelu>(v[u], r[u], b);
  }
  for (; q < n_vec; q += step)
    yv[q] = bias_act_vec<T, kResidual, kRelu>(yv[q], kResidual ? rv[q] : vuint4{0, 0, 0, 0}, b);
}

// General form (channel count not a multiple of the vector width, tiny tensors, tails): scalar.
template <typename T, bool kResidual, bool kRelu>
__global__ void bias_act_scalar_kernel(void *y, const float *bias, const void *res, uint64_t begin, uint64_t n,
                                       uint32_t channels) {
  const uint64_t step = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = begin + static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += step) {
    float o = load_raw<T>(y, i) + bias[i % channels];
    if (kResidual) o += load_raw<T>(res, i);
    if (kRelu) o = o > 0.0f ? o : (o != o ? o : 0.0f);     // NaN goes through, as torch.relu (round 6: it became 0 before)
    if constexpr (std::is_same_v<T, F32>) static_cast<float *>(y)[i] = o;
    else static_cast<uint16_t *>(y)[i] = static_cast<uint16_t>(float_to_storage<T>(o));
  }
}

// ------------------------------------------------------------------------------------------------
// Stem epilogue: bias + ReLU + 3x3 / stride 2 / pad 1 max-pool in ONE pass (ResNet conv1 -> bn1 -> relu
// -> maxpool, reference odtk/backbones/resnet.py:24-39 via torchvision).  Bias-add, ReLU and the
// rounding to the storage type are monotone, so  pool(round(relu(y + b))) == round(relu(pool(y) + b))
// bit for bit: take the maximum of the RAW conv outputs, then apply the epilogue to one value instead
// of nine.  Reads the full-resolution conv output once (neighbouring windows overlap in L2) and
// writes the quarter-size result: the separate epilogue pass (read + write of the biggest activation
// of the network, 262 MB at bs 8) disappears.
// One thread = one output pixel x kPer channels (16 bytes); consecutive threads walk the channels of a
// pixel, then x: every load and store is a coalesced 16-byte access.  Padding never wins (torch pads
// with -inf).  16-bit types only (the inference dtypes); channels % 8 == 0.
// k32 (every real size: fewer than 2^32 output vectors): the output index is split into (image, row, column, channel
// group) by multiply-high (fastdiv.hpp).  Three 64-bit divisions by run-time divisors were ~360 of the ~700 vector
// instructions per output vector of a kernel that the instruction count, not the stream, bounds (round 3).
typedef short pk_short2 __attribute__((ext_vector_type(2)));
// two sign-magnitude 16-bit floats (bf16 or fp16) in a dword -> two order-preserving signed 16-bit keys, and back
__device__ __forceinline__ uint32_t order_keys16(uint32_t w) {
  const pk_short2 sign = __builtin_bit_cast(pk_short2, w) >> 15;                 // v_pk_ashrrev_i16: 0 / -1 per half
  return w ^ (__builtin_bit_cast(uint32_t, sign) & 0x7fff7fffu);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_short2, a), __builtin_bit_cast(pk_short2, b)));
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(pk_short2, a), __builtin_bit_cast(pk_short2, b)));
}

struct PoolDivisors {
  FastDiv groups, wo, ho;
};

template <typename T, bool kRelu, bool k32>
__global__ __launch_bounds__(256) void bias_act_maxpool_kernel(const uint16_t *__restrict__ y, const float *__restrict__ bias,
                                                               uint16_t *__restrict__ out, uint32_t batch, uint32_t h,
                                                               uint32_t w, uint32_t c, uint32_t ho, uint32_t wo,
                                                               const PoolDivisors dv) {
  constexpr int kPer = T::kPerLoad;                        // 8
  const uint32_t groups = c / kPer;
  const uint64_t total = static_cast<uint64_t>(batch) * ho * wo * groups;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += static_cast<uint64_t>(gridDim.x) * 256ull) {
    uint32_t g, ox, oy, b;
    if constexpr (k32) {
      uint32_t p = fastdivmod(static_cast<uint32_t>(i), dv.groups, &g);
      p = fastdivmod(p, dv.wo, &ox);
      b = fastdivmod(p, dv.ho, &oy);
    } else {
      g = static_cast<uint32_t>(i % groups);
      uint64_t p = i / groups;
      ox = static_cast<uint32_t>(p % wo);
      p /= wo;
      oy = static_cast<uint32_t>(p % ho);
      b = static_cast<uint32_t>(p / ho);
    }
    // Round 6: the maximum is taken on the 16-bit patterns themselves, two per instruction.  A sign-magnitude half h becomes an
    // order-preserving two's-complement key by k = h ^ ((h >> 15) & 0x7fff) (an involution); keys of positive NaNs lie above
    // +inf's, keys of negative NaNs below -inf's, so a running packed maximum AND minimum see every NaN of the window: five
    // packed operations per dword of a neighbour where the fp32 form (unpack, two compares, select: per element) took sixteen.
    // (Equal values have equal bits except +0 / -0, where the key order picks +0; the sum with the bias hides the difference
    // unless the bias is -0.)
    uint32_t kmax[4], kmin[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { kmax[d] = 0x80008000u; kmin[d] = 0x7fff7fffu; }
    // Coordinates are clamped instead of tested: a clamped row / column is one the window holds anyway (padding never wins;
    // the maximum ignores duplicates), so the nine loads are unconditional, issued together, and addressed as one 64-bit image
    // base + 32-bit row / column offsets (the host checks h * w * c < 2^32).
    const int y0 = static_cast<int>(oy) * 2 - 1, x0 = static_cast<int>(ox) * 2 - 1;
    const uint16_t *img = y + static_cast<uint64_t>(b) * h * w * c + g * kPer;
    uint32_t ro[3], co[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int yy = y0 + d < 0 ? 0 : (y0 + d >= static_cast<int>(h) ? static_cast<int>(h) - 1 : y0 + d);
      const int xx = x0 + d < 0 ? 0 : (x0 + d >= static_cast<int>(w) ? static_cast<int>(w) - 1 : x0 + d);
      ro[d] = static_cast<uint32_t>(yy) * w * c;
      co[d] = static_cast<uint32_t>(xx) * c;
    }
    vuint4 v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = *reinterpret_cast<const vuint4 *>(img + (ro[dy] + co[dx]));
    }
#pragma unroll
    for (int n9 = 0; n9 < 9; ++n9) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t k = order_keys16(v[n9][d]);
        kmax[d] = pk_max_i16(kmax[d], k);
        kmin[d] = pk_min_i16(kmin[d], k);
      }
    }
    float m[kPer];
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const int32_t kx = static_cast<int16_t>(kmax[e >> 1] >> (16 * (e & 1)));
      const int32_t kn = static_cast<int16_t>(kmin[e >> 1] >> (16 * (e & 1)));
      // below -inf's key (bf16: 0xff80 -> 0x807f, fp16: 0xfc00 -> 0x83ff): a negative NaN -- NaN propagates, as torch's max_pool2d
      constexpr int32_t kNegInfKey = std::is_same_v<T, BF16> ? -32641 : -31745;
      const int32_t kk = kn < kNegInfKey ? kn : kx;
      const uint32_t hx = static_cast<uint32_t>(kk ^ ((kk >> 15) & 0x7fff)) & 0xffffu;
      m[e] = std::is_same_v<T, BF16> ? bf16_bits_to_float(hx) : f16_bits_to_float(hx);
    }
    vuint4 o;
#pragma unroll
    for (int e = 0; e < kPer; e += 2) {
      float a0 = m[e] + bias[g * kPer + e], a1 = m[e + 1] + bias[g * kPer + e + 1];
      if (kRelu) { a0 = a0 > 0.0f ? a0 : (a0 != a0 ? a0 : 0.0f); a1 = a1 > 0.0f ? a1 : (a1 != a1 ? a1 : 0.0f); }
      o[e >> 1] = float_to_storage<T>(a0) | (float_to_storage<T>(a1) << 16);
    }
    *reinterpret_cast<vuint4 *>(out + ((static_cast<uint64_t>(b) * ho + oy) * wo + ox) * c + g * kPer) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Nearest-neighbour 2x upsampling of a channels_last activation: out[b][y][x][c] = in[b][y / 2][x / 2][c] -- the FPN's
// top-down path (reference odtk/backbones/fpn.py:45-61: F.interpolate(p5, scale_factor=2) added to lateral4(c4), and the
// same one level down).  torch's upsample_nearest2d kernel + the channels_last copy behind it ran at 0.65 TB/s (125 us per
// step for 100 MB of traffic, profiles/r03_bench_steady_kernel_stats.txt); this is a plain stream: one thread = one output
// pixel x 16 bytes of channels, the four fine pixels of a coarse one read the same 16 bytes (L2 / TCP hits), every access
// coalesced.  HBM-bound: sizeof(T) x (input + 4 x input) per element.  Any storage type: bytes are copied, never decoded.
__global__ __launch_bounds__(256) void upsample_nearest2x_kernel(const vuint4 *__restrict__ in, vuint4 *__restrict__ out, uint32_t h,
                                                                 uint32_t w, uint32_t groups, uint32_t total, const FastDiv by_groups,
                                                                 const FastDiv by_wo, const FastDiv by_ho) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {   // total < 2^32 - grid (checked by the host)
    uint32_t g, ox, oy;
    uint32_t p = fastdivmod(i, by_groups, &g);
    p = fastdivmod(p, by_wo, &ox);
    const uint32_t b = fastdivmod(p, by_ho, &oy);
    out[i] = in[((static_cast<uint64_t>(b) * h + (oy >> 1)) * w + (ox >> 1)) * groups + g];
  }
}

// ---- stem: space-to-depth pack of the network input ---------------------------------------------------------------------------
// The ResNet stem is a 7x7 / stride-2 convolution over THREE input channels: as an implicit GEMM its reduction runs over
// 3-element channel vectors (MIOpen's pick for it: 230 us at bs 8, a quarter of the matrix-core rate of the other layers, + a
// 32 us helper pass).  A stride-2 convolution over x is a stride-1 convolution over the 2x2 space-to-depth image of x:
//     xs[n][y][x][(dy * 2 + dx) * 3 + c] = x[n][c][2 y + dy][2 x + dx]        12 channels, padded with zeros to 16
//     conv7x7/s2/p3(x, w) == conv4x4/s1/pad(2 before, 1 after)(xs, w4),   w4[k][(dy*2+dx)*3+c][R][S] = w[k][c][2R+dy-1][2S+dx-1]
// (taps outside the 7x7 window are zero) -- the same products, summed in another order, over 16-byte channel vectors.  This kernel
// writes xs in the engine's dtype straight from the caller's image (fp32 / bf16 / fp16, NCHW or channels_last): it replaces the cast
// + layout pass the engine ran on its input anyway.  One thread per output pixel: 2 x 24 contiguous input bytes (channels_last fp32)
// or 6 x 8 (NCHW), 32 contiguous output bytes.  HBM-bound: sizeof(in) * 3 * H * W read, 2 * 16 * H/2 * W/2 written per image.
template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void stem_pack_kernel(const void *x, void *out, uint32_t batch, uint32_t height, uint32_t width,
                                                        uint32_t channels_last, FastDiv by_wo, FastDiv by_howo) {
  const uint32_t ho = height / 2, wo = width / 2;
  const uint64_t total = static_cast<uint64_t>(batch) * ho * wo;
  const bool pairs = width % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0;   // (launch-uniform) rows of pixel pairs are 8-byte aligned
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    uint32_t rem, xo;
    const uint32_t n = fastdivmod(static_cast<uint32_t>(i), by_howo, &rem);   // (batch * ho * wo < 2^32: checked by the host)
    const uint32_t yo = fastdivmod(rem, by_wo, &xo);
    float v[12];
    if (std::is_same_v<TIn, F32> && pairs) {
      // fp32 images (round 6): the two pixels of a row are 8 / 24 contiguous, 8-byte aligned bytes -- six float2 loads, not twelve
      const float *xf = static_cast<const float *>(x) + static_cast<uint64_t>(n) * 3 * height * width;   // (3 * H * W < 2^32: host)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const uint32_t yy = 2 * yo + dy, xx = 2 * xo;
        if (channels_last) {
          const float2 *p = reinterpret_cast<const float2 *>(xf + (yy * width + xx) * 3u);
          const float2 a0 = p[0], a1 = p[1], a2 = p[2];      // (c0 c1) (c2 | c0) (c1 c2) of the pixels dx = 0, 1
          v[(dy * 2) * 3 + 0] = a0.x; v[(dy * 2) * 3 + 1] = a0.y; v[(dy * 2) * 3 + 2] = a1.x;
          v[(dy * 2 + 1) * 3 + 0] = a1.y; v[(dy * 2 + 1) * 3 + 1] = a2.x; v[(dy * 2 + 1) * 3 + 2] = a2.y;
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float2 a0 = *reinterpret_cast<const float2 *>(xf + ((static_cast<uint32_t>(c) * height + yy) * width + xx));
            v[(dy * 2) * 3 + c] = a0.x;
            v[(dy * 2 + 1) * 3 + c] = a0.y;
          }
        }
      }
    } else {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const uint64_t yy = 2 * yo + dy, xx = 2 * xo + dx;
            const uint64_t off = channels_last ? ((static_cast<uint64_t>(n) * height + yy) * width + xx) * 3 + c
                                               : ((static_cast<uint64_t>(n) * 3 + c) * height + yy) * width + xx;
            v[(dy * 2 + dx) * 3 + c] = load_raw<TIn>(x, off);
          }
        }
      }
    }
    vuint4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) o0[e] = float_to_storage<TOut>(v[2 * e]) | (float_to_storage<TOut>(v[2 * e + 1]) << 16);
    o1[0] = float_to_storage<TOut>(v[8]) | (float_to_storage<TOut>(v[9]) << 16);
    o1[1] = float_to_storage<TOut>(v[10]) | (float_to_storage<TOut>(v[11]) << 16);
    o1[2] = 0u;
    o1[3] = 0u;
    vuint4 *dst = reinterpret_cast<vuint4 *>(static_cast<uint16_t *>(out) + i * 16);
    dst[0] = o0;
    dst[1] = o1;
  }
}

}  // namespace odtk

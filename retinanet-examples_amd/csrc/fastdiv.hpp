// fastdiv.hpp -- exact unsigned 32-bit division by a divisor that is fixed for a whole launch.
//
// The streaming kernels turn a flat memory offset into (image, anchor, class, pixel) once per 16-byte vector: three
// divisions by run-time divisors.  gfx950 has no integer divide: `x / d` on two registers is a ~25-instruction
// float-reciprocal + correction sequence, i.e. ~20 VALU slots PER LOGIT on the fp32 loss kernel (4 logits per vector).
// Granlund & Montgomery, "Division by invariant integers using multiplication" (PLDI '94), figure 4.1: with
// l = ceil(log2 d), m' = floor(2^32 (2^l - d) / d) + 1, sh1 = min(l, 1), sh2 = max(l - 1, 0)
//     t = mulhi(m', n);   n / d == (t + ((n - t) >> sh1)) >> sh2          for every n in [0, 2^32), d in [1, 2^32)
// -- five VALU slots (v_mul_hi_u32 is quarter rate on its own), prepared by the host once per launch.
// Plain C++ on both sides (host: launch set-up; device: the kernels); tests/test_abi_host.py checks it against `/`.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define ODTK_HD __host__ __device__ __forceinline__
#else
#define ODTK_HD inline
#endif

namespace odtk {

struct FastDiv {
  uint32_t d;      // the divisor (>= 1)
  uint32_t m;      // m'
  uint32_t sh1;    // 0 for d == 1, otherwise 1
  uint32_t sh2;    // max(ceil(log2 d) - 1, 0)
};

inline FastDiv fastdiv_make(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;                                       // callers validate; never divide by zero in a kernel
  uint32_t l = 0;
  while (l < 32 && (static_cast<uint64_t>(1) << l) < d) ++l;   // ceil(log2 d)
  f.d = d;
  f.m = static_cast<uint32_t>(((static_cast<uint64_t>(1) << 32) * ((static_cast<uint64_t>(1) << l) - d)) / d + 1);
  f.sh1 = l < 1 ? l : 1;
  f.sh2 = l > 1 ? l - 1 : 0;
  return f;
}

ODTK_HD uint32_t fastdiv(uint32_t n, const FastDiv &f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = __umulhi(f.m, n);
#else
  const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(f.m) * n) >> 32);
#endif
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// quotient and remainder in one go
ODTK_HD uint32_t fastdivmod(uint32_t n, const FastDiv &f, uint32_t *rem) {
  const uint32_t q = fastdiv(n, f);
  *rem = n - q * f.d;
  return q;
}

}  // namespace odtk

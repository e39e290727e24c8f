// prelude.hpp -- TEST INFRASTRUCTURE ONLY.  The few CUDA names the reference's device code uses
// (csrc/cuda/nms_iou.cu:41-258, :324-375, csrc/cuda/nms.cu:44-80 and the float6 struct of csrc/cuda/utils.h), provided for a
// plain host C++ compiler so that the reference's OWN rotated-IoU / rotated-NMS source can be compiled
// with g++ and run on the CPU (oracle/ref_build/build_ref.py splices: this file + the reference lines
// read from /root/reference at build time + harness.cpp, piped to g++ -> oracle/_ref/*.so; nothing is written out).
// Built with -ffp-contract=off and without fast-math: this is the IEEE reading of the reference source
// (the reference's own build used nvcc --use_fast_math, whose bits no other compiler reproduces).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <vector>

#define __host__
#define __device__
#define __global__

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

// one emulated thread: the kernels are entered with a 1x1 launch geometry
struct odtk_ref_dim3 { unsigned x, y, z; };
static odtk_ref_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
static inline void __syncthreads() {}

using std::abs;
using std::exp;        // exp(float) must stay a float function, as in device code
using std::isnan;
using std::max;
using std::min;

// what the decode lambdas return: thrust::make_tuple(score, box, class)
template <typename Box> struct odtk_ref_tuple { float score; Box box; int cls; };
namespace thrust {
template <typename Box> static inline odtk_ref_tuple<Box> make_tuple(float s, Box b, int c) { return odtk_ref_tuple<Box>{s, b, c}; }
}

// iou.hpp -- pairwise rotated IoU between ground-truth quads and anchor quads (training-side
// target assignment, odtk/box.py:223).  Replaces iou_cuda_kernel + odtk::cuda::iou
// (csrc/cuda/nms_iou.cu:324-387).  Output layout [num_anchors, num_boxes] row-major -- the net
// effect of the reference's swapped argument order at nms_iou.cu:385 (csrc/extensions.cpp:64-66,
// consumed by `overlap.max(1)` at box.py:226): for pair (anchor i, box j) the ANCHOR quad is the
// subject polygon (padded) and the BOX quad is the clipper.
#pragma once

#include "rotated_iou.hpp"

namespace odtk {

__global__ __launch_bounds__(256) void iou_pairs_kernel(const float *__restrict__ boxes,
                                                        const float *__restrict__ anchors,
                                                        float *__restrict__ out, int num_boxes,
                                                        int num_anchors) {
  __shared__ float2 s_clip[4 * kClipSlotsPerWave];          // lane-private polygon columns (rotated_iou.hpp)
  float2 *clip = s_clip + (threadIdx.x >> 6) * kClipSlotsPerWave + (threadIdx.x & 63);
  const long long pairs = 1ll * num_boxes * num_anchors;
  const long long step = 1ll * gridDim.x * blockDim.x;
  for (long long t = 1ll * blockIdx.x * blockDim.x + threadIdx.x; t < pairs; t += step) {
    const int ai = static_cast<int>(t / num_boxes);
    const int bj = static_cast<int>(t - 1ll * ai * num_boxes);
    Pt I[4], M[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      I[k].x = anchors[ai * 8 + 2 * k];
      I[k].y = anchors[ai * 8 + 2 * k + 1];
      M[k].x = boxes[bj * 8 + 2 * k];
      M[k].y = boxes[bj * 8 + 2 * k + 1];
    }
    out[t] = overlap_from(I, M, clip);
  }
}

}  // namespace odtk

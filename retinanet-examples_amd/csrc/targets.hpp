// targets.hpp -- fused training-target assignment for the pyramid levels of the whole batch
// (SURVEY.md 8f rank 2).  One thread per anchor (image, a, y, x): IoU against every ground-truth box
// of its image (LDS-resident), arg-max, regression deltas, depth and the one-hot class map, written
// straight in the [A, *, H, W] layout the loss consumes.
//
// Replaces the reference's snap_to_anchors (odtk/box.py:134-189), which builds an [A*H*W, N] IoU
// matrix and ~25 small tensor ops per image and level (measured on MI355X: 4.9 ms per training step
// for 5 levels x 2 images = 11-16 % of the step), called from odtk/model.py:167-184.
// Arithmetic follows box.py:150-170 operation by operation (+1 pixel convention, first maximum wins
// like torch.max); only log() differs from the CPU reference by <= 1 ulp.
//
// HBM-bound on the writes: (C + 5) floats per anchor (340 B at C = 80), everything coalesced along x.
#pragma once

#include "common.hpp"
#include "../../include/odtk_hip.h"

namespace odtk {

constexpr int kSnapThreads = 256;
constexpr int kSnapMaxBoxes = 1024;   // target rows staged in LDS at a time (more rows: further rounds)

struct SnapArgs {
  const float *targets;   // [B, n_max, 5] = (x, y, w, h, class), class < 0 marks padding
  float *cls_target;      // [B, A, C, H, W], or null: not wanted (the fused loss derives it from depth)
  float *box_target;      // [B, A, 4, H, W]
  float *depth;           // [B, A, 1, H, W]
  int n_max, num_anchors, num_classes, height, width;
  float stride, iou_bg, iou_fg;
  float anchors[ODTK_MAX_ANCHORS * 4];
};

// One workgroup: 256 consecutive (anchor, y, x) cells of level `a`, image blockIdx.y.
__device__ __forceinline__ void snap_to_anchors_block(const SnapArgs &a, int block_in_level) {
  __shared__ float s_box[kSnapMaxBoxes * 6];   // x1, y1, x2, y2, area, class of the VALID boxes of this round, in order
  __shared__ int s_n;

  const int b = blockIdx.y;
  const float *tg = a.targets + static_cast<size_t>(b) * a.n_max * 5;
  const int hw = a.height * a.width;
  const int cell = block_in_level * kSnapThreads + threadIdx.x;     // (anchor, y, x) flattened
  const bool live = cell < a.num_anchors * hw;
  const int an = live ? cell / hw : 0, pix = live ? cell - an * hw : 0;
  const int y = pix / a.width, x = pix - y * a.width;
  const float gx = static_cast<float>(x) * a.stride, gy = static_cast<float>(y) * a.stride;
  const float ax1 = gx + a.anchors[an * 4 + 0], ay1 = gy + a.anchors[an * 4 + 1];
  const float ax2 = gx + a.anchors[an * 4 + 2], ay2 = gy + a.anchors[an * 4 + 3];
  const float a_area = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);                  // box.py:160

  float best = 0.0f;
  float bq[5] = {0, 0, 0, 0, 0};   // x1, y1, x2, y2, class of the best box so far
  int seen = 0;                    // valid boxes seen in earlier rounds
  for (int base = 0; base < a.n_max || base == 0; base += kSnapMaxBoxes) {
    // compact the valid rows of this round in order (the reference filters `target[target[:, -1] > -1]`): one wave, ballot
    __syncthreads();
    if (threadIdx.x < kWave) {
      int n = 0;
      const int end = base + kSnapMaxBoxes < a.n_max ? base + kSnapMaxBoxes : a.n_max;
      for (int i0 = base; i0 < end; i0 += kWave) {
        const int i = i0 + threadIdx.x;
        float r[5] = {0, 0, 0, 0, -1};
        if (i < end) {
#pragma unroll
          for (int k = 0; k < 5; ++k) r[k] = tg[i * 5 + k];
        }
        const bool valid = r[4] > -1.0f;
        const uint64_t m = __ballot(valid);
        if (valid) {
          const int p = n + __popcll(m & ((1ull << threadIdx.x) - 1ull));
          const float x2 = r[0] + r[2] - 1.0f, y2 = r[1] + r[3] - 1.0f;          // box.py:155
          s_box[p * 6 + 0] = r[0]; s_box[p * 6 + 1] = r[1]; s_box[p * 6 + 2] = x2; s_box[p * 6 + 3] = y2;
          s_box[p * 6 + 4] = (x2 - r[0] + 1.0f) * (y2 - r[1] + 1.0f);              // box.py:159
          s_box[p * 6 + 5] = r[4];
        }
        n += __popcll(m);
      }
      if (threadIdx.x == 0) s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int i = 0; i < n; ++i) {
      const float *q = s_box + i * 6;
      float w = tmin_nan(ax2, q[2]) - tmax_nan(ax1, q[0]) + 1.0f;               // box.py:156-158
      float h = tmin_nan(ay2, q[3]) - tmax_nan(ay1, q[1]) + 1.0f;
      w = w < 0.0f ? 0.0f : w;
      h = h < 0.0f ? 0.0f : h;
      const float inter = w * h;
      const float ov = inter / (a_area + q[4] - inter);                          // box.py:161
      if ((seen == 0 && i == 0) || ov > best || (ov != ov && best == best)) {    // first max; NaN wins like torch.max
        best = ov;
        bq[0] = q[0]; bq[1] = q[1]; bq[2] = q[2]; bq[3] = q[3]; bq[4] = q[5];
      }
    }
    seen += n;
  }
  if (!live) return;
  const size_t img = static_cast<size_t>(b) * a.num_anchors;
  float *box = a.box_target + (img + an) * 4 * hw + pix;
  float *dep = a.depth + (img + an) * hw + pix;

  int hot = -1;                 // class channel that gets the 1 (none: background / no boxes)
  float dl[4] = {0, 0, 0, 0}, dval = 0.0f;
  if (seen > 0) {
    // box2delta (box.py:67-78)
    const float aw = ax2 - ax1 + 1.0f, ah = ay2 - ay1 + 1.0f;
    const float acx = ax1 + 0.5f * aw, acy = ay1 + 0.5f * ah;
    const float bw = bq[2] - bq[0] + 1.0f, bh = bq[3] - bq[1] + 1.0f;
    const float bcx = bq[0] + 0.5f * bw, bcy = bq[1] + 0.5f * bh;
    dl[0] = (bcx - acx) / aw;
    dl[1] = (bcy - acy) / ah;
    dl[2] = logf(bw / aw);
    dl[3] = logf(bh / ah);
    const bool bg = best < a.iou_bg, fg = best >= a.iou_fg;
    dval = fg ? bq[4] + 1.0f : (bg ? 0.0f : -1.0f);                              // box.py:173-175
    if (!bg) hot = static_cast<int>(bq[4]);                                      // box.py:179-184 (.long())
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) box[static_cast<size_t>(k) * hw] = dl[k];
  *dep = dval;
  if (a.cls_target) {
    float *cls = a.cls_target + (img + an) * a.num_classes * hw + pix;
    for (int c = 0; c < a.num_classes; ++c) cls[static_cast<size_t>(c) * hw] = (c == hot) ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(kSnapThreads) void snap_to_anchors_kernel(const SnapArgs a) {
  snap_to_anchors_block(a, static_cast<int>(blockIdx.x));
}

// All pyramid levels of the batch in ONE launch (a level table in the kernel arguments, like the loss): five launches of
// 2 .. 170 workgroups each cost 44 us per training step, mostly launch boundaries.
struct SnapLevelsArgs {
  SnapArgs lv[ODTK_MAX_LEVELS];
  uint32_t block_begin[ODTK_MAX_LEVELS + 1];
  int n_levels;
};
static_assert(sizeof(SnapLevelsArgs) <= 4096, "kernel arguments travel by value");

__global__ __launch_bounds__(kSnapThreads) void snap_to_anchors_levels_kernel(const SnapLevelsArgs a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < a.n_levels && blockIdx.x >= a.block_begin[i]) l = i;
  snap_to_anchors_block(a.lv[l], static_cast<int>(blockIdx.x - a.block_begin[l]));
}

}  // namespace odtk

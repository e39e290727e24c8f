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
#include "rotated_iou.hpp"
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

// ------------------------------------------------------------------------------------------------
// Rotated boxes (reference odtk/box.py:192-252 + csrc/cuda/nms_iou.cu:324-387): the same assignment with the polygon IoU
// of the ground-truth quad and the rotated anchor quad as the overlap.  The reference -- and rounds 1-3 here -- built the
// full [27 * H * W, N] IoU matrix with one kernel (432 000 x N polygon clips at P3) and ~15 torch ops per image and level
// behind it.  Here one thread owns one anchor cell of one image: IoU against the image's boxes (LDS-resident: axis form,
// quad, class, bounding box), arg-max, deltas + (sin, cos), depth, class map, all levels of the batch in ONE launch.
// The clip (~1000 wave instructions per 64 pairs) only runs for pairs whose bounding boxes come within reach of each other:
// two quads further apart along x or y than 2 px + the rounding allowance cannot intersect, the clip would return an
// empty polygon and the reference's value is exactly 0 / union = +0 -- nearly every (anchor, box) pair of a level.
// (Condition for skipping: the box's quad has edges of at least one pixel and the union term is finite and positive;
// anything else takes the full path.)  Arithmetic of the overlap = rotated_iou.hpp (bit-equal to the reference's device code
// compiled for the CPU, tests/test_gpu_rotated.py); deltas as box.py:81-94.
constexpr int kSnapRotBoxes = 128;    // boxes staged in LDS at a time (20 floats each)

struct SnapRotArgs {
  const float *gt_axis;        // [B, n_max, 6] = x1, y1, x2, y2, sin, cos   (utils.rotate_boxes: x, y, x + w - 1, y + h - 1)
  const float *gt_quads;       // [B, n_max, 8] ordered corners
  const float *gt_class;       // [B, n_max], < 0 marks padding
  const float *anchors_axis;   // DEVICE [A, 4]
  const float *anchors_rot;    // DEVICE [A, 8]
  float *cls_target;           // [B, A, C, H, W] or null
  float *box_target;           // [B, A, 6, H, W]
  float *depth;                // [B, A, 1, H, W]
  int n_max, num_anchors, num_classes, height, width;
  float stride, iou_bg, iou_fg;
};

struct SnapRotLevelsArgs {
  SnapRotArgs lv[ODTK_MAX_LEVELS];
  uint32_t block_begin[ODTK_MAX_LEVELS + 1];
  int n_levels;
};
static_assert(sizeof(SnapRotLevelsArgs) <= 4096, "kernel arguments travel by value");

__global__ __launch_bounds__(kSnapThreads) void snap_to_anchors_rotated_levels_kernel(const SnapRotLevelsArgs args) {
  __shared__ float s_box[kSnapRotBoxes * 20];   // per valid box: axis[6], quad[8], class, bbox x0 y0 x1 y1, ok-to-skip flag
  __shared__ int s_n;
  __shared__ float2 s_clip[(kSnapThreads / kWave) * kClipSlotsPerWave];
  float2 *clip = s_clip + (threadIdx.x >> 6) * kClipSlotsPerWave + (threadIdx.x & 63);

  int l = 0;
#pragma unroll
  for (int i = 1; i < ODTK_MAX_LEVELS; ++i)
    if (i < args.n_levels && blockIdx.x >= args.block_begin[i]) l = i;
  const SnapRotArgs &a = args.lv[l];
  const int block_in_level = static_cast<int>(blockIdx.x - args.block_begin[l]);
  const int b = blockIdx.y;
  const int hw = a.height * a.width;
  const int cell = block_in_level * kSnapThreads + threadIdx.x;     // (anchor, y, x) flattened
  const bool live = cell < a.num_anchors * hw;
  const int an = live ? cell / hw : 0, pix = live ? cell - an * hw : 0;
  const int y = pix / a.width, x = pix - y * a.width;
  const float gx = static_cast<float>(x) * a.stride, gy = static_cast<float>(y) * a.stride;
  // the anchor: axis form and quad, grid + anchor like the reference's meshgrid sum (box.py:213-218)
  float ax[4];
  Pt I[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    ax[k] = ((k & 1) ? gy : gx) + a.anchors_axis[an * 4 + k];
    I[k].x = gx + a.anchors_rot[an * 8 + 2 * k];
    I[k].y = gy + a.anchors_rot[an * 8 + 2 * k + 1];
  }
  float ix0 = I[0].x, ix1 = I[0].x, iy0 = I[0].y, iy1 = I[0].y;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    ix0 = fminf(ix0, I[k].x); ix1 = fmaxf(ix1, I[k].x);
    iy0 = fminf(iy0, I[k].y); iy1 = fmaxf(iy1, I[k].y);
  }
  const float i_area2 = fabsf(quad_shoelace(I));

  float best = 0.0f;
  float bq[7] = {0, 0, 0, 0, 0, 0, 0};   // axis form + class of the best box so far
  int seen = 0;
  for (int base = 0; base < a.n_max || base == 0; base += kSnapRotBoxes) {
    __syncthreads();
    if (threadIdx.x < kWave) {           // compact the valid rows of this round in order (one wave, ballot)
      int n = 0;
      const int end = base + kSnapRotBoxes < a.n_max ? base + kSnapRotBoxes : a.n_max;
      for (int i0 = base; i0 < end; i0 += kWave) {
        const int i = i0 + threadIdx.x;
        const size_t row = static_cast<size_t>(b) * a.n_max + i;
        const float cls = i < end ? a.gt_class[row] : -1.0f;
        const bool valid = cls > -1.0f;
        const uint64_t m = __ballot(valid);
        if (valid) {
          float *o = s_box + (n + __popcll(m & ((1ull << threadIdx.x) - 1ull))) * 20;
#pragma unroll
          for (int k = 0; k < 6; ++k) o[k] = a.gt_axis[row * 6 + k];
          float q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { q[k] = a.gt_quads[row * 8 + k]; o[6 + k] = q[k]; }
          o[14] = cls;
          float x0 = q[0], x1 = q[0], y0 = q[1], y1 = q[1], shortest = 3.0e38f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            x0 = fminf(x0, q[2 * k]); x1 = fmaxf(x1, q[2 * k]);
            y0 = fminf(y0, q[2 * k + 1]); y1 = fmaxf(y1, q[2 * k + 1]);
            const float ex = q[2 * ((k + 1) & 3)] - q[2 * k], ey = q[2 * ((k + 1) & 3) + 1] - q[2 * k + 1];
            shortest = fminf(shortest, ex * ex + ey * ey);
          }
          o[15] = x0; o[16] = y0; o[17] = x1; o[18] = y1;
          // may pairs with this box be skipped on distance?  edges >= 1 px, finite coordinates (NaN fails every test)
          o[19] = (shortest >= 1.0f && x1 - x0 < 3.0e38f && y1 - y0 < 3.0e38f) ? 1.0f : 0.0f;
        }
        n += __popcll(m);
      }
      if (threadIdx.x == 0) s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    for (int i = 0; i < n; ++i) {
      const float *q = s_box + i * 20;
      Pt M[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { M[k].x = q[6 + 2 * k]; M[k].y = q[6 + 2 * k + 1]; }
      // distance reject: exactly what the clip would return (+0) for quads that cannot touch
      const float gap = fmaxf(fmaxf(ix0 - q[17], q[15] - ix1), fmaxf(iy0 - q[18], q[16] - iy1));
      const float reach = fmaxf(fmaxf(fabsf(ix0), fabsf(ix1)), fmaxf(fabsf(iy0), fabsf(iy1))) +
                          fmaxf(fmaxf(fabsf(q[15]), fabsf(q[17])), fmaxf(fabsf(q[16]), fabsf(q[18])));
      const float uni2 = i_area2 + fabsf(quad_shoelace(M));
      float ov;
      if (q[19] != 0.0f && gap > 2.0f + 1e-3f * reach + 4e-6f * reach * reach && uni2 > 0.0f && uni2 < 3.0e38f) ov = 0.0f;
      else ov = overlap_from(I, M, clip);
      if ((seen == 0 && i == 0) || ov > best || (ov != ov && best == best)) {    // first max; NaN wins like torch.max
        best = ov;
#pragma unroll
        for (int k = 0; k < 6; ++k) bq[k] = q[k];
        bq[6] = q[14];
      }
    }
    seen += n;
  }
  if (!live) return;
  const size_t img = static_cast<size_t>(b) * a.num_anchors;
  float *box = a.box_target + (img + an) * 6 * hw + pix;
  float *dep = a.depth + (img + an) * hw + pix;
  int hot = -1;
  float dl[6] = {0, 0, 0, 0, 0, 0}, dval = 0.0f;
  if (seen > 0) {
    // box2delta_rotated (box.py:67-94): box2delta on the axis forms, (sin, cos) passed through
    const float aw = ax[2] - ax[0] + 1.0f, ah = ax[3] - ax[1] + 1.0f;
    const float acx = ax[0] + 0.5f * aw, acy = ax[1] + 0.5f * ah;
    const float bw = bq[2] - bq[0] + 1.0f, bh = bq[3] - bq[1] + 1.0f;
    const float bcx = bq[0] + 0.5f * bw, bcy = bq[1] + 0.5f * bh;
    dl[0] = (bcx - acx) / aw;
    dl[1] = (bcy - acy) / ah;
    dl[2] = logf(bw / aw);
    dl[3] = logf(bh / ah);
    dl[4] = bq[4];
    dl[5] = bq[5];
    const bool bg = best < a.iou_bg, fg = best >= a.iou_fg;
    dval = fg ? bq[6] + 1.0f : (bg ? 0.0f : -1.0f);                              // box.py:233-235
    if (!bg) hot = static_cast<int>(bq[6]);                                      // box.py:238-247 (.long())
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) box[static_cast<size_t>(k) * hw] = dl[k];
  *dep = dval;
  if (a.cls_target) {
    float *cls = a.cls_target + (img + an) * a.num_classes * hw + pix;
    for (int c = 0; c < a.num_classes; ++c) cls[static_cast<size_t>(c) * hw] = (c == hot) ? 1.0f : 0.0f;
  }
}

}  // namespace odtk

// rotated_iou.hpp -- rotated-rectangle IoU by convex polygon clipping, gfx950 device code.
//
// Behavioural contract = the reference's CUDA implementation (the reference has no runnable CPU
// rotated path): csrc/cuda/nms_iou.cu:114-169 (clip of a <=8-gon by the four edges of the other
// quad, shoelace area), :199-248 (quad construction from [x1,y1,x2,y2,sin,cos], the 0.001 pad on
// coinciding same-numbered corners, "union" = area_i + area_m, NaN rules).  Arithmetic is plain
// IEEE fp32 in the written order (no FMA contraction, IEEE division) so that the CPU restatement
// oracle/c/odtk_oracle.c gives the same bits; the reference's own build used --use_fast_math
// (setup.py:15), whose bits cannot be reproduced anywhere else.
//
// Deviation (documented): the reference writes past its 8-entry vertex arrays when a
// degenerate clip emits more than 8 vertices, and calls rotateLeft(.., 0) on an empty polygon
// (nms_iou.cu:139-149, :127-128) -- undefined behaviour.  Here a polygon is capped at 8 vertices
// (extra vertices are dropped) and an empty polygon stays empty.
#pragma once

#include "common.hpp"

namespace odtk {

constexpr int kPolyMax = 8;

struct Pt { float x, y; };

// Clips polygon P (count vertices) against the 4 directed edges of quad R; returns |area|.
__device__ __forceinline__ float clip_area(const Pt *R, Pt *P) {
  int count = 4;
#pragma unroll 1
  for (int e = 0; e < 4; ++e) {
    const Pt r1 = R[e], r2 = R[(e + 1) & 3];
    // Line(r1, r2): a = r2.y - r1.y, b = r1.x - r2.x, c = r2 x r1      (nms_iou.cu:86)
    const float la = r2.y - r1.y, lb = r1.x - r2.x, lc = r2.x * r1.y - r2.y * r1.x;
    float lv[kPolyMax];
#pragma unroll
    for (int j = 0; j < kPolyMax; ++j) lv[j] = (j < count) ? (la * P[j].x + lb * P[j].y + lc) : 0.0f;
    Pt Q[kPolyMax];
    int nq = 0;
#pragma unroll 1
    for (int j = 0; j < count; ++j) {
      const int jn = (j + 1 == count) ? 0 : j + 1;
      if (lv[j] <= 0.0f) { if (nq < kPolyMax) Q[nq] = P[j]; ++nq; }
      if (lv[j] * lv[jn] <= 0.0f) {
        // Line(P[j], P[jn]) intersected with the clip line                (nms_iou.cu:92-95)
        const Pt r3 = P[j], r4 = P[jn];
        const float ma = r4.y - r3.y, mb = r3.x - r4.x, mc = r4.x * r3.y - r4.y * r3.x;
        const float w = la * mb - lb * ma;
        Pt x;
        x.x = (lb * mc - lc * mb) / w;
        x.y = (lc * ma - la * mc) / w;
        if (nq < kPolyMax) Q[nq] = x;
        ++nq;
      }
    }
    count = nq < kPolyMax ? nq : kPolyMax;
#pragma unroll
    for (int j = 0; j < kPolyMax; ++j) if (j < count) P[j] = Q[j];
  }
  float area = 0.0f;
  if (count > 2) {
#pragma unroll 1
    for (int k = 0; k < count; ++k) {
      const int kn = (k + 1 == count) ? 0 : k + 1;
      area += P[k].x * P[kn].y - P[k].y * P[kn].x;
    }
  }
  return fabsf(area / 2.0f);
}

__device__ __forceinline__ float quad_shoelace(const Pt *R) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += R[k].x * R[(k + 1) & 3].y - R[k].y * R[(k + 1) & 3].x;
  return s;
}

// corners of [x1,y1,x2,y2] rotated about the box centre by (cos, sin)   (nms_iou.cu:199-228)
__device__ __forceinline__ void rotated_corners(const float *b, float s, float c, Pt *out) {
  const float cx = (b[0] + b[2]) / 2.0f, cy = (b[1] + b[3]) / 2.0f;
  const float dx[4] = {b[0] - cx, b[2] - cx, b[2] - cx, b[0] - cx};
  const float dy[4] = {b[1] - cy, b[1] - cy, b[3] - cy, b[3] - cy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    out[k].x = (dx[k] * c - dy[k] * s) + cx;
    out[k].y = (dy[k] * c + dx[k] * s) + cy;
  }
}

// overlap with the reference's NaN rules (nms_iou.cu:240-247)
__device__ __forceinline__ float overlap_from(const Pt *I, const Pt *M) {
  Pt P[kPolyMax];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    P[k].x = I[k].x + (I[k].x == M[k].x ? 0.001f : 0.0f);    // pad coinciding same-numbered corners
    P[k].y = I[k].y + (I[k].y == M[k].y ? 0.001f : 0.0f);
  }
#pragma unroll
  for (int k = 4; k < kPolyMax; ++k) { P[k].x = 0.0f; P[k].y = 0.0f; }
  const float inter = clip_area(M, P);
  const float uni = (fabsf(quad_shoelace(I)) + fabsf(quad_shoelace(M))) / 2.0f;
  if (inter != inter && uni != uni) return 1.0f;
  if (inter != inter) return 0.0f;
  return inter / (uni - inter);
}

// Does the kept box m suppress the lower-scored box j?  boxes are [x1,y1,x2,y2,sin,cos].
// Reference default (nms_iou.cu:186-193): BOTH quads are rotated by j's (sin, cos).
__device__ __forceinline__ bool rotated_suppresses(const float *m, const float *j, float thr, bool own_angle) {
  Pt I[4], M[4];
  rotated_corners(j, j[4], j[5], I);
  rotated_corners(m, own_angle ? m[4] : j[4], own_angle ? m[5] : j[5], M);
  return overlap_from(I, M) > thr;
}

}  // namespace odtk

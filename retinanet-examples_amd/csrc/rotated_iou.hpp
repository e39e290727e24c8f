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
//
// Register plan.  The reference keeps ~7 float2[8] arrays per thread in local memory; a direct
// port spills to scratch (176 B/lane) and every IoU pays dozens of dependent memory round trips
// (measured: ~2.5 us per IoU inside the NMS).  Here the current polygon P[8] lives in VGPRs: every
// loop over its vertices is fully unrolled and predicated on `j < count`, so all READS are
// statically indexed.  Only the emitted polygon is written at a per-lane varying position; those
// writes go to a lane-private LDS column (slot k of lane l at q[k*64 + l], conflict-free) and are
// read back with static slots.  No scratch.
#pragma once

#include "common.hpp"

namespace odtk {

constexpr int kPolyMax = 8;
constexpr int kClipSlotsPerWave = kPolyMax * kWave;      // float2 slots of LDS per wave (4 KiB)

struct Pt { float x, y; };

// Clips polygon P (4 vertices on entry, in registers) against the 4 directed edges of quad R;
// returns |area|.  q = this lane's column in a kClipSlotsPerWave-sized LDS region of its wave.
__device__ __forceinline__ float clip_area(const Pt *R, Pt *P, float2 *q) {
  int count = 4;
  // the four clip edges run through ONE copy of the code: the quad's corners rotate through four registers pairs (static
  // indices, so they stay registers) instead of the loop being unrolled four times -- a quarter of the code, and the
  // scheduler cannot overlap two edges' temporaries (the unrolled form needed 74 VGPRs alone and spilled inside the NMS)
  Pt Q0 = R[0], Q1 = R[1], Q2 = R[2], Q3 = R[3];
#pragma unroll 1
  for (int e = 0; e < 4; ++e) {
    const Pt r1 = Q0, r2 = Q1;
    { const Pt t = Q0; Q0 = Q1; Q1 = Q2; Q2 = Q3; Q3 = t; }
    // Line(r1, r2): a = r2.y - r1.y, b = r1.x - r2.x, c = r2 x r1      (nms_iou.cu:86)
    const float la = r2.y - r1.y, lb = r1.x - r2.x, lc = r2.x * r1.y - r2.y * r1.x;
    float lv[kPolyMax];
#pragma unroll
    for (int j = 0; j < kPolyMax; ++j) lv[j] = la * P[j].x + lb * P[j].y + lc;        // only j < count is used
    int nq = 0;
#pragma unroll
    for (int j = 0; j < kPolyMax; ++j) {
      if (j < count) {
        const bool wrap = (j + 1 == count);
        const Pt pn = wrap ? P[0] : P[(j + 1) & (kPolyMax - 1)];                        // next vertex, static index
        const float lvn = wrap ? lv[0] : lv[(j + 1) & (kPolyMax - 1)];
        if (lv[j] <= 0.0f) {                                                            // nms_iou.cu:140-143
          if (nq < kPolyMax) q[nq * kWave] = make_float2(P[j].x, P[j].y);
          ++nq;
        }
        if (lv[j] * lvn <= 0.0f) {                                                      // :144-150
          // Line(P[j], P[j+1]) intersected with the clip line                         (:92-95)
          const float ma = pn.y - P[j].y, mb = P[j].x - pn.x, mc = pn.x * P[j].y - pn.y * P[j].x;
          const float w = la * mb - lb * ma;
          const float ix = (lb * mc - lc * mb) / w, iy = (lc * ma - la * mc) / w;
          if (nq < kPolyMax) q[nq * kWave] = make_float2(ix, iy);
          ++nq;
        }
      }
    }
    count = nq < kPolyMax ? nq : kPolyMax;
#pragma unroll
    for (int j = 0; j < kPolyMax; ++j) {                    // statically indexed read-back
      const float2 v = q[j * kWave];
      P[j].x = v.x;
      P[j].y = v.y;
    }
  }
  float area = 0.0f;
  if (count > 2) {
#pragma unroll
    for (int k = 0; k < kPolyMax; ++k) {
      const Pt pn = (k + 1 == count) ? P[0] : P[(k + 1) & (kPolyMax - 1)];
      const float term = P[k].x * pn.y - P[k].y * pn.x;                                 // :163-165
      area = (k < count) ? area + term : area;
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
__device__ __forceinline__ float overlap_from(const Pt *I, const Pt *M, float2 *q) {
  Pt P[kPolyMax];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    P[k].x = I[k].x + (I[k].x == M[k].x ? 0.001f : 0.0f);    // pad coinciding same-numbered corners
    P[k].y = I[k].y + (I[k].y == M[k].y ? 0.001f : 0.0f);
  }
#pragma unroll
  for (int k = 4; k < kPolyMax; ++k) { P[k].x = 0.0f; P[k].y = 0.0f; }
  const float uni = (fabsf(quad_shoelace(I)) + fabsf(quad_shoelace(M))) / 2.0f;   // (before the clip: I need not stay in registers)
  const float inter = clip_area(M, P, q);
  if (inter != inter && uni != uni) return 1.0f;
  if (inter != inter) return 0.0f;
  return inter / (uni - inter);
}

// Cheap reject in front of the polygon clip: true = the kept box m CANNOT suppress the lower-scored box j.
// Every corner of a quad lies within its half diagonal (scaled by |(sin, cos)|, which the network does not normalise) of
// the box centre, so two quads whose centres are further apart -- along x or along y -- than an upper bound of the two
// radii cannot intersect: the clip would return an empty polygon, overlap 0, and `0 > thr` is false for thr >= 0.  The
// margin (0.1 % + 2 px + 4e-6 * coordinate^2) is far above the fp32 error of the clip's line equations as long as the
// kept quad is properly oriented with edges of at least one pixel (otherwise, and for any NaN / inf, the answer is "cannot
// tell" = false and the full path decides).  In an NMS almost all same-class pairs are far apart.
__device__ __forceinline__ bool rotated_far_apart(const float *m, const float *j, float thr, bool own_angle) {
  if (!(thr >= 0.0f)) return false;
  const float sm = own_angle ? m[4] : j[4], cm = own_angle ? m[5] : j[5];
  const float wm = m[2] - m[0], hm = m[3] - m[1];
  const float kj = fabsf(j[4]) + fabsf(j[5]), km = fabsf(sm) + fabsf(cm);           // >= |(sin, cos)|: no sqrt needed
  const float rr = 0.5f * ((fabsf(j[2] - j[0]) + fabsf(j[3] - j[1])) * kj + (fabsf(wm) + fabsf(hm)) * km);   // >= r_j + r_m
  const float sxm = m[0] + m[2], sym = m[1] + m[3];                                 // twice the kept centre
  const float far = 0.5f * fmaxf(fabsf((j[0] + j[2]) - sxm), fabsf((j[1] + j[3]) - sym));
  const float reach = 0.5f * (fabsf(sxm) + fabsf(sym)) + far + rr;                  // >= |any coordinate| of both quads
  // |(sin, cos)| >= (|sin| + |cos|) / sqrt 2: edges of the kept quad are at least one pixel long
  const bool proper = wm * km >= 1.5f && hm * km >= 1.5f;                          // implies wm, hm > 0 (km >= 0)
  return proper && far > rr * 1.001f + 2.0f + 4e-6f * reach * reach;
}

// Does the kept box m suppress the lower-scored box j?  boxes are [x1,y1,x2,y2,sin,cos].
// Reference default (nms_iou.cu:186-193): BOTH quads are rotated by j's (sin, cos).
__device__ __forceinline__ bool rotated_suppresses(const float *m, const float *j, float thr, bool own_angle, float2 *q) {
  Pt I[4], M[4];
  rotated_corners(j, j[4], j[5], I);
  rotated_corners(m, own_angle ? m[4] : j[4], own_angle ? m[5] : j[5], M);
  return overlap_from(I, M, q) > thr;
}

}  // namespace odtk

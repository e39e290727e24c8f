/*
 * odtk_oracle.c -- plain-C CPU restatement of the reference's post-processing algorithms.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): built by oracle/c/Makefile, loaded through
 * ctypes by oracle/c_oracle.py, used by tests/ as a checker.  Never linked into the product.
 *
 * Pinning: the rotated IoU / rotated NMS below reproduce, bit for bit, the reference's own nms_iou.cu device
 * code compiled for the CPU (oracle/ref_build, tests/test_oracle_native_ref.py); decode / axis nms are
 * tied to the torch restatement that is pinned to the reference's box.py (tests/test_oracle_c.py).
 *
 * What is restated, and from where (paths relative to /root/reference):
 *   decode (axis + rotated) : odtk/box.py:255-309 (CPU conventions: `>=` threshold, two-sided
 *                             clamp box.py:105-111, score-descending output) with the gather /
 *                             6-tuple layout of csrc/cuda/decode_rotate.cu:116-167.
 *   nms (axis)              : odtk/box.py:312-367.
 *   rotated IoU             : csrc/cuda/nms_iou.cu:114-169 (IntersectionArea), :199-248 (quad
 *                             construction, 0.001 pad, area_i + area_m "union", NaN rules).
 *   nms (rotated)           : csrc/cuda/nms_iou.cu:171-258 (greedy structure, quirk at :192 --
 *                             both quads rotated by the lower-scored box's angle) with the CPU
 *                             path's ordering / stop-after-ndetections (box.py:402-425).
 *   pairwise iou            : csrc/cuda/nms_iou.cu:324-387 incl. the [num_anchors, num_boxes]
 *                             layout that results from the swapped call at :385.
 *
 * Arithmetic: IEEE fp32, every operation in the written order (-ffp-contract=off, no fast-math).
 * exp() in decode is the correctly rounded fp32 exp ((float)exp((double)x)), which is what the
 * HIP kernel computes; torch's CPU expf (the reference's) differs from it by <= 1 ulp on ~1 % of
 * inputs, so boxes from THIS file are compared bit-for-bit with the HIP path while boxes from
 * oracle/box_oracle.py (torch arithmetic, pinned bit-exact to the reference) are compared within
 * tolerance.  Ordering is the canonical stable rule (score desc, index asc).
 *
 * Undefined behaviour in the reference that is NOT reproduced (documented deviation): a clip that
 * emits more than 8 vertices writes past the reference's float2[8] arrays (nms_iou.cu:139-149) and
 * an empty polygon makes rotateLeft(.., 0) write array[-1] (:127-128).  Here polygons are capped
 * at 8 vertices and an empty polygon stays empty.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* ordering helpers                                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float score; int64_t index; } cand_t;

static int cand_cmp(const void *pa, const void *pb) {
  const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
  if (a->score > b->score) return -1;          /* score descending */
  if (a->score < b->score) return 1;
  return (a->index > b->index) - (a->index < b->index);   /* then index ascending */
}

static float exp_cr(float x) { return (float)exp((double)x); }

/* torch.max(m, torch.min(t, M)) with NaN propagation (box.py:107) */
static float clamp_like_torch(float t, float hi) {
  float mn = (t != t) ? t : (t < hi ? t : hi);
  return (mn != mn) ? mn : (mn > 0.0f ? mn : 0.0f);
}

/* ------------------------------------------------------------------------------------------ */
/* decode                                                                                      */
/* ------------------------------------------------------------------------------------------ */
/* cls [B, A*C, H, W], box [B, A*nb, H, W] (nb = 4 or 6), anchors [A, 4].
 * outputs: scores [B, top_n], boxes [B, top_n, nb], classes [B, top_n], indices [B, top_n] (-1 pad) */
int oracle_decode(const float *cls, const float *box, int B, int A, int C, int H, int W, int stride,
                  const float *anchors, float threshold, int top_n, int nb,
                  float *out_scores, float *out_boxes, float *out_classes, int64_t *out_indices) {
  const int64_t hw = (int64_t)H * W, n = (int64_t)A * C * hw;
  cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
  if (!cand) return -1;
  const float fstride = (float)stride;
  const float lim_x = (float)W * fstride - 1.0f, lim_y = (float)H * fstride - 1.0f;
  for (int b = 0; b < B; ++b) {
    const float *s = cls + (int64_t)b * n;
    const float *d = box + (int64_t)b * A * nb * hw;
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
      if (s[i] >= threshold) { cand[k].score = s[i]; cand[k].index = i; ++k; }   /* box.py:283 */
    qsort(cand, (size_t)k, sizeof(cand_t), cand_cmp);
    const int64_t keep = k < top_n ? k : top_n;
    for (int64_t t = 0; t < top_n; ++t) {
      float *ob = out_boxes + ((int64_t)b * top_n + t) * nb;
      if (t >= keep) {
        out_scores[(int64_t)b * top_n + t] = 0.0f;
        out_classes[(int64_t)b * top_n + t] = 0.0f;
        out_indices[(int64_t)b * top_n + t] = -1;
        for (int c = 0; c < nb; ++c) ob[c] = 0.0f;
        continue;
      }
      const int64_t i = cand[t].index;
      const int64_t x = i % W, y = (i / W) % H, c = (i / hw) % C, a = i / (hw * C);   /* box.py:291-297 */
      float dl[6];
      for (int q = 0; q < nb; ++q) dl[q] = d[((int64_t)a * nb + q) * hw + y * W + x];
      const float fx = (float)x * fstride, fy = (float)y * fstride;                  /* box.py:302 */
      const float ax1 = fx + anchors[4 * a], ay1 = fy + anchors[4 * a + 1];
      const float ax2 = fx + anchors[4 * a + 2], ay2 = fy + anchors[4 * a + 3];
      const float w = ax2 - ax1 + 1.0f, h = ay2 - ay1 + 1.0f;                        /* box.py:100 */
      const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
      const float pcx = dl[0] * w + cx, pcy = dl[1] * h + cy;
      const float pw = exp_cr(dl[2]) * w, ph = exp_cr(dl[3]) * h;
      ob[0] = clamp_like_torch(pcx - 0.5f * pw, lim_x);
      ob[1] = clamp_like_torch(pcy - 0.5f * ph, lim_y);
      ob[2] = clamp_like_torch(pcx + 0.5f * pw - 1.0f, lim_x);
      ob[3] = clamp_like_torch(pcy + 0.5f * ph - 1.0f, lim_y);
      if (nb == 6) { ob[4] = dl[4]; ob[5] = dl[5]; }                                 /* decode_rotate.cu:152-162 */
      out_scores[(int64_t)b * top_n + t] = cand[t].score;
      out_classes[(int64_t)b * top_n + t] = (float)c;
      out_indices[(int64_t)b * top_n + t] = i;
    }
  }
  free(cand);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* rotated IoU                                                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y; } pt_t;
#define POLY_MAX 8

/* clip polygon P (4 vertices on entry) by the 4 directed edges of quad R; |shoelace|/2 */
static float clip_area(const pt_t *R, pt_t *P) {
  int count = 4;
  for (int e = 0; e < 4; ++e) {
    const pt_t r1 = R[e], r2 = R[(e + 1) & 3];
    const float la = r2.y - r1.y, lb = r1.x - r2.x, lc = r2.x * r1.y - r2.y * r1.x;   /* nms_iou.cu:86 */
    float lv[POLY_MAX];
    for (int j = 0; j < count; ++j) lv[j] = la * P[j].x + lb * P[j].y + lc;             /* :90 */
    pt_t Q[POLY_MAX];
    int nq = 0;
    for (int j = 0; j < count; ++j) {
      const int jn = (j + 1 == count) ? 0 : j + 1;
      if (lv[j] <= 0.0f) { if (nq < POLY_MAX) Q[nq] = P[j]; ++nq; }                     /* :140-143 */
      if (lv[j] * lv[jn] <= 0.0f) {                                                     /* :144-150 */
        const pt_t r3 = P[j], r4 = P[jn];
        const float ma = r4.y - r3.y, mb = r3.x - r4.x, mc = r4.x * r3.y - r4.y * r3.x;
        const float w = la * mb - lb * ma;                                              /* :93 */
        pt_t x;
        x.x = (lb * mc - lc * mb) / w;
        x.y = (lc * ma - la * mc) / w;
        if (nq < POLY_MAX) Q[nq] = x;
        ++nq;
      }
    }
    count = nq < POLY_MAX ? nq : POLY_MAX;
    for (int j = 0; j < count; ++j) P[j] = Q[j];
  }
  float area = 0.0f;
  if (count > 2)
    for (int k = 0; k < count; ++k) {
      const int kn = (k + 1 == count) ? 0 : k + 1;
      area += P[k].x * P[kn].y - P[k].y * P[kn].x;                                      /* :163-165 */
    }
  return fabsf(area / 2.0f);
}

static float quad_shoelace(const pt_t *R) {
  float s = 0.0f;
  for (int k = 0; k < 4; ++k) s += R[k].x * R[(k + 1) & 3].y - R[k].y * R[(k + 1) & 3].x;
  return s;
}

static void rotated_corners(const float *b, float s, float c, pt_t *out) {              /* :199-228 */
  const float cx = (b[0] + b[2]) / 2.0f, cy = (b[1] + b[3]) / 2.0f;
  const float dx[4] = {b[0] - cx, b[2] - cx, b[2] - cx, b[0] - cx};
  const float dy[4] = {b[1] - cy, b[1] - cy, b[3] - cy, b[3] - cy};
  for (int k = 0; k < 4; ++k) {
    out[k].x = (dx[k] * c - dy[k] * s) + cx;
    out[k].y = (dy[k] * c + dx[k] * s) + cy;
  }
}

static float overlap_from(const pt_t *I, const pt_t *M) {                                /* :233-247 */
  pt_t P[POLY_MAX];
  for (int k = 0; k < 4; ++k) {
    P[k].x = I[k].x + (I[k].x == M[k].x ? 0.001f : 0.0f);
    P[k].y = I[k].y + (I[k].y == M[k].y ? 0.001f : 0.0f);
  }
  const float inter = clip_area(M, P);
  const float uni = (fabsf(quad_shoelace(I)) + fabsf(quad_shoelace(M))) / 2.0f;
  if (inter != inter && uni != uni) return 1.0f;
  if (inter != inter) return 0.0f;
  return inter / (uni - inter);
}

/* boxes [N, 8] corner quads, anchors [M, 8]; out [M, N]  (anchor = subject, box = clipper) */
void oracle_iou_pairs(const float *boxes, const float *anchors, int N, int M, float *out) {
  for (int ai = 0; ai < M; ++ai)
    for (int bj = 0; bj < N; ++bj) {
      pt_t I[4], Mq[4];
      for (int k = 0; k < 4; ++k) {
        I[k].x = anchors[ai * 8 + 2 * k]; I[k].y = anchors[ai * 8 + 2 * k + 1];
        Mq[k].x = boxes[bj * 8 + 2 * k]; Mq[k].y = boxes[bj * 8 + 2 * k + 1];
      }
      out[(int64_t)ai * N + bj] = overlap_from(I, Mq);
    }
}

/* overlap of two [x1,y1,x2,y2,sin,cos] boxes as the rotated NMS computes it: j = lower-scored
 * ("i" in the reference), m = kept box; own_angle = 0 reproduces nms_iou.cu:186-193. */
float oracle_rotated_overlap(const float *m, const float *j, int own_angle) {
  pt_t I[4], Mq[4];
  rotated_corners(j, j[4], j[5], I);
  rotated_corners(m, own_angle ? m[4] : j[4], own_angle ? m[5] : j[5], Mq);
  return overlap_from(I, Mq);
}

/* ------------------------------------------------------------------------------------------ */
/* nms (axis nb = 4, rotated nb = 6)                                                           */
/* ------------------------------------------------------------------------------------------ */
static float tmaxf(float a, float b) { return (a > b || a != a) ? a : b; }
static float tminf(float a, float b) { return (a < b || a != a) ? a : b; }

int oracle_nms(const float *scores, const float *boxes, const float *classes, int B, int count, int nb,
               float thresh, int ndet, int own_angle,
               float *out_scores, float *out_boxes, float *out_classes, int64_t *out_indices) {
  cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(count > 0 ? count : 1));
  unsigned char *dead = (unsigned char *)malloc((size_t)(count > 0 ? count : 1));
  if (!cand || !dead) { free(cand); free(dead); return -1; }
  for (int b = 0; b < B; ++b) {
    const float *s = scores + (int64_t)b * count, *bx = boxes + (int64_t)b * count * nb;
    const float *cl = classes + (int64_t)b * count;
    int k = 0;
    for (int i = 0; i < count; ++i)
      if (s[i] > 0.0f) { cand[k].score = s[i]; cand[k].index = i; ++k; }               /* box.py:328 */
    qsort(cand, (size_t)k, sizeof(cand_t), cand_cmp);
    memset(dead, 0, (size_t)(k > 0 ? k : 1));
    int kept = 0;
    for (int m = 0; m < k && kept < ndet; ++m) {
      if (dead[m]) continue;
      const int mi = (int)cand[m].index;
      const float *mb = bx + (int64_t)mi * nb;
      const int64_t o = (int64_t)b * ndet + kept;
      out_scores[o] = cand[m].score;
      out_classes[o] = cl[mi];
      out_indices[o] = mi;
      for (int c = 0; c < nb; ++c) out_boxes[o * nb + c] = mb[c];
      ++kept;
      if (kept == ndet) break;
      const float marea = (mb[2] - mb[0] + 1.0f) * (mb[3] - mb[1] + 1.0f);               /* box.py:339 */
      for (int j = m + 1; j < k; ++j) {
        if (dead[j]) continue;
        const int ji = (int)cand[j].index;
        if (cl[ji] != cl[mi]) continue;                                                  /* box.py:351 */
        const float *jb = bx + (int64_t)ji * nb;
        int suppress;
        if (nb == 4) {
          const float x1 = tmaxf(jb[0], mb[0]), y1 = tmaxf(jb[1], mb[1]);               /* box.py:346-348 */
          const float x2 = tminf(jb[2], mb[2]), y2 = tminf(jb[3], mb[3]);
          float w = x2 - x1 + 1.0f, h = y2 - y1 + 1.0f;
          w = w < 0.0f ? 0.0f : w;
          h = h < 0.0f ? 0.0f : h;
          const float inter = w * h;
          const float jarea = (jb[2] - jb[0] + 1.0f) * (jb[3] - jb[1] + 1.0f);
          const float iou = inter / (jarea + marea - inter);                             /* box.py:350 */
          suppress = !(iou <= thresh);
        } else {
          suppress = oracle_rotated_overlap(mb, jb, own_angle) > thresh;                 /* nms_iou.cu:248 */
        }
        if (suppress) dead[j] = 1;
      }
    }
    for (int t = kept; t < ndet; ++t) {
      const int64_t o = (int64_t)b * ndet + t;
      out_scores[o] = 0.0f; out_classes[o] = 0.0f; out_indices[o] = -1;
      for (int c = 0; c < nb; ++c) out_boxes[o * nb + c] = 0.0f;
    }
  }
  free(cand);
  free(dead);
  return 0;
}

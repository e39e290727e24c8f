/*
 * odtk_hip.h -- C ABI of the MI355X (gfx950) post-processing library  libodtk_hip.so
 *
 * Drop-in boundary for the per-anchor hot path of NVIDIA/retinanet-examples (ODTK):
 * box decode + threshold/top-k prefilter, batched class-aware NMS, and the rotated-bbox
 * decode / IoU / NMS variants.  Every entry point is `extern "C"`, takes plain pointers and
 * sizes, never allocates device memory, never synchronises the host with the device, never
 * throws, and only ENQUEUES work on the caller's HIP stream.
 *
 * Conventions shared by all entry points
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - all tensor pointers are DEVICE pointers; `anchors` is a HOST pointer (copied by value
 *     into the kernel arguments -- nothing is uploaded, nothing outlives the call).
 *   - two-phase workspace query, cub style, exactly like the reference
 *     (csrc/cuda/decode.cu:53-72, nms.cu:87-105): call with workspace == NULL or
 *     workspace_size == 0 -> the return value is the number of scratch bytes required
 *     (> 0); call again with a buffer of at least that size -> returns ODTK_OK (0).
 *   - errors are negative return values (the reference never checked a CUDA error and threw
 *     std::runtime_error("Workspace is too small!") across the boundary, utils.h:55-57).
 *   - the workspace is scratch: contents before the call are irrelevant (nothing is cleared by a launch of
 *     its own: the first kernel of a call writes every word the second reads), a call may not
 *     share its workspace with another call that is in flight on a different stream.
 *   - outputs are fully written (zero padded tails); they need not be pre-zeroed.
 *
 * Semantics are those of the reference's CPU path odtk/box.py (the normative ones, see
 * DESIGN.md): candidates are `score >= score_thresh`, ordered score-descending with ties
 * broken by ascending flat NCHW index (decode) / ascending candidate position (nms);
 * boxes are clamped on both sides to [0, size*stride-1]; NMS uses the +1 pixel convention
 * and suppresses iff same class and not (IoU <= nms_thresh).
 */
#ifndef ODTK_HIP_H
#define ODTK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODTK_OK                0
#define ODTK_ERR_INVALID      -1   /* bad argument (null pointer, size 0, limit exceeded) */
#define ODTK_ERR_WORKSPACE    -2   /* workspace smaller than the size query reports       */
#define ODTK_ERR_HIP          -3   /* a HIP runtime call failed; see odtk_last_hip_error() */
#define ODTK_ERR_UNSUPPORTED  -4   /* dtype / layout combination not implemented          */

#define ODTK_MAX_LEVELS    6       /* pyramid levels per call (P3..P7 = 5); keeps kernargs < 4 KiB */
#define ODTK_MAX_ANCHORS   32      /* anchors per cell (9 axis-aligned, 27 rotated)       */
#define ODTK_MAX_TOP_N     16384   /* per-level top_n (<= 4096: 32 KiB LDS sort; beyond: a 128 KiB variant) */
#define ODTK_MAX_NMS_COUNT 7680    /* candidates per image nms keeps LDS-resident (5 x 1000 by default);  */
#define ODTK_MAX_NMS_COUNT_SCRATCH (1 << 22) /* beyond that, up to this, the key list lives in the workspace   */
#define ODTK_MAX_NMS_DETECTIONS 2048 /* detections_per_im (100 by default)                */

/* element types of the head tensors */
#define ODTK_F32   0
#define ODTK_BF16  1
#define ODTK_F16   2

/* flags */
#define ODTK_FLAG_ROTATED        1u   /* 6-parameter boxes [x1,y1,x2,y2,sin,cos]                       */
#define ODTK_FLAG_LOGITS         2u   /* cls holds logits: sigmoid is fused into the prefilter          */
#define ODTK_FLAG_ROTATED_NMS_FIXED_ANGLE 4u /* rotated nms: use each box's OWN angle (the reference   */
                                      /* rotates both quads by the lower-scored box's angle,            */
                                      /* csrc/cuda/nms_iou.cu:192; that is the default here too)        */

const char *odtk_version(void);
/* ABI guard.  sizeof() of a struct type of this header AS THE LIBRARY WAS COMPILED: which = 0 odtk_level_t,
 * 1 odtk_snap_level_t, 2 odtk_snap_rot_level_t, 3 odtk_loss_level_t; -1 for any other value.  A binding compiled (or
 * mirrored) against another revision of this header would pass level arrays of the wrong stride: both bindings compare
 * their own sizeof with this at load time and refuse to load on a mismatch (odtk/_C.py, csrc/odtk_binding.cpp). */
int odtk_abi_struct_size(int which);
/* hipGetErrorString of the last HIP failure seen by this thread ("" if none). */
const char *odtk_last_hip_error(void);

/*
 * odtk_decode / odtk_decode_rotate -- one pyramid level, whole batch.
 * Replaces  int odtk::cuda::decode(...)         csrc/cuda/decode.h:30-35  (decode.cu:44-171)
 *      and  int odtk::cuda::decode_rotate(...)  csrc/cuda/decode_rotate.h:30-35
 * Same argument order and meaning; `const std::vector<float>& anchors` becomes
 * (anchors, anchors_len) and cudaStream_t becomes a hipStream_t.
 *   inputs[0]  scores  float32 [batch, num_anchors*num_classes, height, width]  contiguous NCHW
 *   inputs[1]  deltas  float32 [batch, num_anchors*{4|6},       height, width]
 *   outputs[0] scores  float32 [batch, top_n]
 *   outputs[1] boxes   float32 [batch, top_n, {4|6}]
 *   outputs[2] classes float32 [batch, top_n]
 *   anchors    host float[anchors_len], anchors_len == 4*num_anchors (axis-aligned set,
 *              also for rotated -- csrc/cuda/decode_rotate.cu:139, box.py:258-259)
 */
int odtk_decode(int batch_size, const void *const *inputs, void *const *outputs,
                size_t height, size_t width, size_t scale, size_t num_anchors, size_t num_classes,
                const float *anchors, size_t anchors_len, float score_thresh, int top_n,
                void *workspace, size_t workspace_size, void *stream);

int odtk_decode_rotate(int batch_size, const void *const *inputs, void *const *outputs,
                       size_t height, size_t width, size_t scale, size_t num_anchors, size_t num_classes,
                       const float *anchors, size_t anchors_len, float score_thresh, int top_n,
                       void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_nms / odtk_nms_rotate -- whole batch, one workgroup per image.
 * Replaces  int odtk::cuda::nms(...)         csrc/cuda/nms.h:28-31      (nms.cu:82-160)
 *      and  int odtk::cuda::nms_rotate(...)  csrc/cuda/nms_iou.h:28-31  (nms_iou.cu:260-322)
 *   inputs[0]  scores  float32 [batch, count]          (<= 0 entries are padding)
 *   inputs[1]  boxes   float32 [batch, count, {4|6}]
 *   inputs[2]  classes float32 [batch, count]
 *   outputs[0..2]      float32 [batch, detections_per_im], [.., {4|6}], [..]
 * Workspace: ALWAYS query (two-phase convention).  Axis-aligned with count <= ODTK_MAX_NMS_COUNT: everything is
 * LDS-resident and the query returns a token size -- unless detections_per_im is in the thousands and the kept list
 * leaves no room, then, as for larger counts (the reference has no cap, nms.cu:82-160), batch * count * 8 bytes for
 * the key lists.  Rotated: additionally the first round in NMS order and its pairwise suppression matrix
 * (m x m bits per image, m = min(count, 8 * detections_per_im, 1024) rounded up to 64): the rotated NMS is three to
 * five launches (first round -> matrix on the whole chip -> resolve; csrc/nms.hpp).
 */
int odtk_nms(int batch_size, const void *const *inputs, void *const *outputs,
             size_t count, int detections_per_im, float nms_thresh,
             void *workspace, size_t workspace_size, void *stream);

int odtk_nms_rotate(int batch_size, const void *const *inputs, void *const *outputs,
                    size_t count, int detections_per_im, float nms_thresh,
                    void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_iou -- pairwise rotated-rectangle IoU (training-side target assignment).
 * Replaces  int odtk::cuda::iou(...)  csrc/cuda/nms_iou.h:33-35 (nms_iou.cu:324-387).
 *   inputs[0]  boxes   float32 [num_boxes, 4 corners, 2]
 *   inputs[1]  anchors float32 [num_anchors, 4 corners, 2]
 *   outputs[0] iou     float32 [num_anchors, num_boxes]   (layout of csrc/extensions.cpp:64-66)
 */
int odtk_iou(const void *const *inputs, void *const *outputs, int num_boxes, int num_anchors,
             void *stream);

/* ------------------------------------------------------------------------------------------
 * Extended entry points (no reference equivalent): the MI355X-first batched forms.
 * ---------------------------------------------------------------------------------------- */

typedef struct odtk_level {
  const void *cls;          /* [batch, A*C, H, W] scores (or logits with ODTK_FLAG_LOGITS)   */
  const void *box;          /* [batch, A*nb, H, W] deltas                                    */
  int32_t height, width;    /* H, W of this level                                            */
  int32_t stride;           /* pixels per cell (the reference's `scale`)                     */
  int32_t channels_last;    /* 0: NCHW-contiguous, 1: NHWC-contiguous (torch.channels_last)  */
  const float *anchors;     /* HOST float[4*A]                                               */
  /* Optional per-channel bias of the heads' last convolutions, folded into the kernels so that the
   * caller can skip its bias pass over the largest activation of the network (both may be NULL):
   *   cls_bias  DEVICE float32 [A*C]: logit = float(cls) + cls_bias[channel] in fp32, then the sigmoid.
   *             Needs ODTK_FLAG_LOGITS, a 16-bit dtype, channels_last = 1 and A*C a multiple of 8;
   *             otherwise ODTK_ERR_UNSUPPORTED.
   *   box_bias  DEVICE float32 [A*nb]: delta = float(box) + box_bias[a*nb + k] in fp32. */
  const float *cls_bias;
  const float *box_bias;
  /* Optional (may be NULL), with cls_bias: the prefilter's per-channel threshold table for THIS cls_bias, score_thresh and
   * dtype, prepared once by odtk_prefilter_thresholds (an engine does it when it folds its weights).  Without it every
   * workgroup of the prefilter derives the table from cls_bias before its first load.  The table carries what it was made
   * for; one made for another threshold / dtype / channel count is ignored.  A table made from OTHER bias values is the
   * caller's bug (remake it whenever cls_bias changes). */
  const float *cls_thresholds;
} odtk_level_t;

/* odtk_prefilter_thresholds -- fills `table` (DEVICE, 16-byte aligned, ODTK_THRESHOLD_TABLE_FLOATS(channels) floats) for
 * odtk_level_t::cls_thresholds.  cls_bias: DEVICE float32 [channels]; dtype: ODTK_BF16 / ODTK_F16; channels % 8 == 0. */
#define ODTK_THRESHOLD_TABLE_FLOATS(channels) ((size_t)(channels) + 8)
int odtk_prefilter_thresholds(const float *cls_bias, int channels, int dtype, float score_thresh, float *table, void *stream);

/*
 * odtk_decode_levels -- ALL pyramid levels x whole batch in one enqueue (2 kernel launches: prefilter,
 * select + decode; no host synchronisation; the reference needs >= 5 launches + 1 host sync per image per
 * level, decode.cu:86-168).  Outputs are written directly in the concatenated layout that
 * `torch.cat(per_level, 1)` produces in the reference (odtk/model.py:164):
 *   outputs[0] scores  float32 [batch, n_levels*top_n]
 *   outputs[1] boxes   float32 [batch, n_levels*top_n, {4|6}]
 *   outputs[2] classes float32 [batch, n_levels*top_n]
 *   outputs[3] (optional, may be NULL) int32 [batch, n_levels*top_n] flat NCHW index of each
 *              detection inside its level (-1 padding) -- for index-level parity checks.
 * n_outputs is 3 or 4.
 */
int odtk_decode_levels(int batch_size, int n_levels, const odtk_level_t *levels,
                       int num_anchors, int num_classes, int dtype, uint32_t flags,
                       float score_thresh, int top_n,
                       void *const *outputs, int n_outputs,
                       void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_nms_ex -- odtk_nms / odtk_nms_rotate selected by flags, plus an optional 4th output
 *   outputs[3] (may be NULL) int32 [batch, detections_per_im]: input position of each kept box.
 */
int odtk_nms_ex(int batch_size, const void *const *inputs, void *const *outputs, int n_outputs,
                size_t count, int detections_per_im, float nms_thresh, uint32_t flags,
                void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_nms_sorted_runs -- odtk_nms_ex for a caller who KNOWS how its candidates are ordered: `count` = n_runs (<= 8) runs of
 * `run_len` candidates, each run sorted by (score desc, position asc) with its non-positive scores at the end -- what decode
 * writes per level and `torch.cat(per_level, 1)` lays side by side (reference odtk/model.py:153-164) -- and run_valid DEVICE
 * uint32 [batch, n_runs] = the entries with a positive score at the head of every run.  The kernel then takes its rounds as
 * prefixes of the runs (no compaction, no radix selection, no sort: csrc/nms.hpp "sorted runs"); odtk_detect calls the same
 * code with what its own decode wrote.  Same outputs as odtk_nms_ex on the same candidates; a run that is NOT sorted is the
 * caller's bug.  (Also what tools/nms_clustered_probe.py replays a trained detector's candidates through.)
 */
int odtk_nms_sorted_runs(int batch_size, const void *const *inputs, void *const *outputs, int n_outputs, size_t count,
                         int run_len, const uint32_t *run_valid, int detections_per_im, float nms_thresh, uint32_t flags,
                         void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_detect -- decode_levels + nms back to back on one stream (the whole post-processing of
 * odtk/model.py:153-165 in 3 launches).  outputs as odtk_nms; workspace holds the candidates.
 */
int odtk_detect(int batch_size, int n_levels, const odtk_level_t *levels,
                int num_anchors, int num_classes, int dtype, uint32_t flags,
                float score_thresh, int top_n, float nms_thresh, int detections_per_im,
                void *const *outputs, void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_bias_act -- fused convolution epilogue, in place, on a channels_last (NHWC) activation:
 *     y[p][c] = act( y[p][c] + bias[c] (+ residual[p][c]) ),   act = ReLU if relu != 0
 * y, residual: device [n_pixels, channels] of `dtype` (ODTK_F32/BF16/F16), 16-byte aligned;
 * bias: DEVICE float32 [channels].  No reference kernel equivalent: it replaces the separate
 * conv-bias / frozen-batch-norm / residual-add / ReLU passes of the reference's PyTorch graph
 * (odtk/backbones/layers.py:5-16, torchvision Bottleneck.forward, odtk/model.py:57-62) once the
 * frozen BN scale is folded into the convolution weights (see odtk/fused.py).
 */
int odtk_bias_act(void *y, const float *bias, const void *residual, size_t n_pixels, int channels,
                  int dtype, int relu, void *stream);

/*
 * odtk_bias_act_maxpool -- stem epilogue: out = maxpool3x3/s2/p1( act( y + bias[c] ) ) in one pass over a
 * channels_last activation (ResNet conv1 -> bn1 -> relu -> maxpool, reference odtk/backbones/resnet.py
 * via torchvision's ResNet.forward).  y: device [batch, height, width, channels], out: device
 * [batch, (height+1)/2, (width+1)/2, channels], both ODTK_BF16 or ODTK_F16, 16-byte aligned,
 * channels % 8 == 0; bias: DEVICE float32 [channels].  Bit-identical to odtk_bias_act followed by
 * torch's max_pool2d (the epilogue is monotone, so it commutes with the maximum).
 */
int odtk_bias_act_maxpool(const void *y, const float *bias, void *out, int batch_size, int height, int width,
                          int channels, int dtype, int relu, void *stream);

/*
 * odtk_upsample_nearest2x -- out[b][y][x][c] = x[b][y / 2][x / 2][c] for a channels_last activation (the FPN's top-down path,
 * reference odtk/backbones/fpn.py:45-61: F.interpolate(., scale_factor=2)).  x: device [batch, height, width, channels], out:
 * device [batch, 2*height, 2*width, channels], both of `dtype`, 16-byte aligned, channels * sizeof(dtype) a multiple of 16.
 * A plain stream (bytes are copied, never decoded): bit-identical to torch's nearest interpolation.
 */
int odtk_upsample_nearest2x(const void *x, void *out, int batch_size, int height, int width, int channels, int dtype,
                            void *stream);

/*
 * odtk_stem_pack -- 2x2 space-to-depth pack of the network input for the ResNet stem, with the cast to the engine's dtype:
 *     out[n][y][x][(dy * 2 + dx) * 3 + c] = x[n][c][2 y + dy][2 x + dx]   for the 12 real channels, channels 12..15 = 0
 * x: device [batch, 3, height, width] of in_dtype (ODTK_F32 / BF16 / F16), NCHW-contiguous (channels_last = 0) or NHWC-contiguous
 * (channels_last = 1); height and width even.  out: device [batch, height / 2, width / 2, 16] of out_dtype (ODTK_BF16 / F16),
 * 16-byte aligned.  The 7x7 / stride-2 / pad-3 stem convolution over x equals a 4x4 / stride-1 convolution with padding (2 before,
 * 1 after) over `out` with the re-indexed weights w4[k][(dy*2+dx)*3+c][R][S] = w[k][c][2R+dy-1][2S+dx-1] (zero outside the 7x7
 * window): same products, 16-byte channel vectors instead of 3-element ones (odtk/fused.py).  Replaces the cast + layout pass of
 * the input; no reference kernel equivalent (the reference feeds torchvision's conv1, odtk/backbones/resnet.py).
 */
int odtk_stem_pack(const void *x, void *out, int batch_size, int height, int width, int in_dtype, int channels_last, int out_dtype,
                   void *stream);

/*
 * odtk_gemm_bias_act -- 1x1 (pointwise) convolution of a channels_last activation as ONE GEMM with
 * the whole epilogue fused:
 *     y[p][o] = act( sum_c x[p][c] * w[o][c] + bias[o] (+ residual[p][o]) ),   act = ReLU if relu != 0
 * x: device [m, k], w: device [n, k] (= the convolution weight [C_out, C_in, 1, 1]), y / residual:
 * device [m, n], all of `dtype` (ODTK_BF16/F16/F32), row-major, 16-byte aligned; bias: DEVICE float32
 * [n]; residual may be NULL and must not alias y.  The contraction runs on hipBLASLt (bound at run time,
 * see odtk_gemm_init); the first call for a shape times the library's candidate kernels on `stream`
 * (synchronises once), later calls only enqueue.  workspace: device scratch for the library (32 MiB is
 * plenty), caller-owned like every other buffer of this ABI.
 * Like odtk_bias_act it has no reference kernel equivalent: it replaces conv1x1 -> frozen BN ->
 * (+ skip) -> ReLU of the reference's PyTorch graph (torchvision Bottleneck.forward,
 * odtk/backbones/layers.py:5-16, odtk/backbones/fpn.py lateral convs) once BN is folded.
 * Returns ODTK_ERR_UNSUPPORTED when hipBLASLt cannot be bound or has no kernel for the problem.
 *
 * odtk_gemm_init -- bind hipBLASLt from `path` (NULL: the already mapped / default libhipblaslt.so.1).
 * Optional; odtk_gemm_bias_act binds the default on first use.  Inside a PyTorch process pass the
 * copy PyTorch ships (torch/lib/libhipblaslt.so) so that only one copy of the soname is in use.
 */
int odtk_gemm_init(const char *hipblaslt_path);
/*
 * Reproducible plans.  Which hipBLASLt solution a problem runs on is decided by a stopwatch at its first call; the choice is not
 * the same on every box, and two solutions add up in different orders.  odtk_gemm_plan_export writes one line per tuned problem
 * ("gemm m n k dtype relu residual solution-index"; returns the bytes the text needs, NUL included, and writes at most
 * `capacity`), odtk_gemm_plan_import takes such lines (others are ignored; returns how many it took): a problem first seen after
 * the import runs on the named solution without timing, if the library still offers it for the problem (otherwise it is timed as
 * usual and counted by odtk_gemm_plan_pin_misses).  The engine's plan file (odtk/fused.py: plan_state / load_plan, ODTK_CONV_PLAN)
 * carries these lines next to libodtk_conv.so's (include/odtk_conv.h).  No reference equivalent (one deterministic PyTorch graph,
 * odtk/model.py:125-165).
 */
size_t odtk_gemm_plan_export(char *text, size_t capacity);
int odtk_gemm_plan_import(const char *text);
int odtk_gemm_plan_pin_misses(void);
int odtk_gemm_bias_act(void *y, const void *x, const void *w, const float *bias, const void *residual,
                       size_t m, int n, int k, int dtype, int relu,
                       void *workspace, size_t workspace_size, void *stream);

/*
 * odtk_snap_to_anchors -- fused training-target assignment for ONE pyramid level of the whole batch.
 * Replaces the reference's pure-torch snap_to_anchors (odtk/box.py:134-189, called per image and
 * level from odtk/model.py:167-184): IoU of every anchor against every ground-truth box (+1 pixel
 * convention), first maximum wins, regression deltas (box2delta, box.py:67-78), depth
 * (-1 ignore / 0 background / class+1) and the one-hot class map.
 *   targets     float32 [batch, n_max, 5] = (x, y, w, h, class); rows with class < 0 are padding
 *   anchors     HOST float[4*num_anchors]
 *   cls_target  float32 [batch, A, C, H, W]   box_target [batch, A, 4, H, W]   depth [batch, A, 1, H, W]
 * Any n_max (rows go through LDS 1024 at a time).  cls_target may be NULL when the caller feeds
 * odtk_retina_loss_* (which derives the class target from depth) and does not need the one-hot map.
 */
int odtk_snap_to_anchors(int batch_size, const float *targets, int n_max, const float *anchors,
                         int num_anchors, int num_classes, int height, int width, int stride,
                         float iou_background, float iou_foreground,
                         float *cls_target, float *box_target, float *depth, void *stream);

/* ... and all pyramid levels in ONE launch (a level table in the kernel arguments; n_levels <= ODTK_MAX_LEVELS). */
typedef struct odtk_snap_level {
  const float *anchors;       /* HOST float[4*num_anchors] of this level's stride */
  float *cls_target;          /* [batch, A, C, H, W] or NULL */
  float *box_target;          /* [batch, A, 4, H, W] */
  float *depth;               /* [batch, A, 1, H, W] */
  int32_t height, width, stride, pad_;
} odtk_snap_level_t;
int odtk_snap_to_anchors_levels(int batch_size, const float *targets, int n_max, int n_levels,
                                const odtk_snap_level_t *levels, int num_anchors, int num_classes,
                                float iou_background, float iou_foreground, void *stream);

/*
 * odtk_snap_to_anchors_rotated_levels -- the same assignment for ROTATED boxes (reference odtk/box.py:192-252, whose overlap is
 * the pairwise polygon IoU csrc/cuda/nms_iou.cu:324-387), all pyramid levels of the batch in ONE launch: no [27*H*W, N] IoU
 * matrix, no per-image host loop.  The ground truth arrives as the reference's `rotate_boxes` leaves it (utils.py:33-82):
 *   gt_axis   float32 [batch, n_max, 6] = (x1, y1, x2, y2, sin, cos)      gt_quads  float32 [batch, n_max, 8] ordered corners
 *   gt_class  float32 [batch, n_max], rows with class < 0 are padding
 * and per level DEVICE tables of the anchors (generate_anchors_rotated: axis form [A, 4] and quads [A, 8]):
 *   cls_target [batch, A, C, H, W] (or NULL)   box_target [batch, A, 6, H, W]   depth [batch, A, 1, H, W]
 */
typedef struct odtk_snap_rot_level {
  const float *anchors_axis;  /* DEVICE float[4*num_anchors] */
  const float *anchors_quads; /* DEVICE float[8*num_anchors] */
  float *cls_target;          /* or NULL */
  float *box_target;
  float *depth;
  int32_t height, width, stride, pad_;
} odtk_snap_rot_level_t;
int odtk_snap_to_anchors_rotated_levels(int batch_size, const float *gt_axis, const float *gt_quads, const float *gt_class,
                                        int n_max, int n_levels, const odtk_snap_rot_level_t *levels, int num_anchors,
                                        int num_classes, float iou_background, float iou_foreground, void *stream);

/*
 * odtk_retina_loss_forward / odtk_retina_loss_backward -- fused, masked FocalLoss + SmoothL1 reduction of ONE
 * pyramid level for the whole batch (the step after target assignment in training).
 * Replaces, per level, reference odtk/model.py:193-209 + odtk/loss.py:13-31 (focal loss, smooth-L1, the two
 * masks, the three sums: ~20 elementwise torch kernels over the classification head and their autograd twins).
 *   cls         [batch, A*C, H, W] logits          dtype ODTK_F32 / BF16 / F16, NCHW or channels_last,
 *   box         [batch, A*box_params, H, W]        same dtype and layout as cls, 16-byte aligned
 *   depth       float32 [batch, A, 1, H, W]        -1 ignore / 0 background / class + 1 (odtk_snap_to_anchors)
 *   box_target  float32 [batch, A, box_params, H, W]
 * The one-hot class target is not an input: under the mask `depth >= 0` it equals `c == depth - 1`.
 * forward:  sums = DEVICE double[3], overwritten with { sum of masked focal losses, sum of masked smooth-L1
 *           losses, number of foreground anchors (depth > 0) }.
 * backward: grad_cls_sum / grad_box_sum = DEVICE float32 scalars (upstream gradients of the two sums; NULL = 0);
 *           dcls / dbox receive d(out)/d(cls) and d(out)/d(box) in the dtype and layout of cls / box.
 * Nothing is read back to the host.  batch*A*C*H*W must be < 2^32.
 */
int odtk_retina_loss_forward(const void *cls, const void *box, const float *depth, const float *box_target,
                             int batch_size, int num_anchors, int num_classes, int height, int width,
                             int box_params, int dtype, int channels_last, float alpha, float gamma, float beta,
                             double *sums, void *stream);
int odtk_retina_loss_backward(const void *cls, const void *box, const float *depth, const float *box_target,
                              int batch_size, int num_anchors, int num_classes, int height, int width,
                              int box_params, int dtype, int channels_last, float alpha, float gamma, float beta,
                              const float *grad_cls_sum, const float *grad_box_sum, void *dcls, void *dbox,
                              void *stream);

/*
 * odtk_retina_loss_levels_forward / _backward -- the same reduction for ALL pyramid levels of the batch in ONE launch
 * per direction (a training step: two launches instead of ten).  Per level: the tensors of odtk_retina_loss_*; every
 * level shares batch, num_anchors, num_classes, box_params, dtype.
 *   forward:  sums = DEVICE double[n_levels][3] (cls_sum, box_sum, foreground count per level), overwritten.
 *   backward: grad_cls_sums / grad_box_sums = DEVICE float32 [n_levels] (NULL = zeros); levels[l].dcls / .dbox receive
 *             the gradients (forward ignores them).
 */
typedef struct odtk_loss_level {
  const void *cls;            /* [batch, A*C, H, W] logits                                   */
  const void *box;            /* [batch, A*box_params, H, W]                                 */
  const float *depth;         /* float32 [batch, A, 1, H, W]                                 */
  const float *box_target;    /* float32 [batch, A, box_params, H, W]                        */
  void *dcls, *dbox;          /* backward outputs, dtype / layout of cls / box               */
  int32_t height, width;
  int32_t channels_last;      /* layout of cls and box: 0 NCHW, 1 NHWC                       */
  int32_t pad_;
} odtk_loss_level_t;

int odtk_retina_loss_levels_forward(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                    int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                    double *sums, void *stream);
/* The forward through a workspace: every workgroup writes its three sums to the workspace and a second (tiny) launch adds
 * them up in a fixed order -- no atomics (the plain form above ends every workgroup in a double atomic on its level's
 * word, which bounds how many workgroups it can afford: DESIGN.md section 4) and a result that is bitwise reproducible
 * from run to run.  Two-phase: workspace == NULL returns the bytes needed for these shapes; `sums` is overwritten. */
int odtk_retina_loss_levels_forward_ws(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                       int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                       double *sums, void *workspace, size_t workspace_size, void *stream);
int odtk_retina_loss_levels_backward(int n_levels, const odtk_loss_level_t *levels, int batch_size, int num_anchors,
                                     int num_classes, int box_params, int dtype, float alpha, float gamma, float beta,
                                     const float *grad_cls_sums, const float *grad_box_sums, void *stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (used by bench.py; off by default, zero cost when off).
 * While enabled, a kernel launch of this library carries a hipEvent pair (asynchronous -- still no
 * host synchronisation): the post-processing, target and loss kernels hand the pair to the launch
 * itself (hipExtLaunchKernelGGL), so the two events hold the dispatch's own begin / end timestamps --
 * the figures rocprofv3's kernel trace reports -- and so do, since round 4, the epilogue kernels (bias_act, the stem's
 * pool pass, the upsampling); only odtk_gemm_bias_act, whose kernels hipBLASLt launches itself, records the pair around
 * the call (marker packets: the dispatch latency is inside the figure on both sides).  odtk_profile_collect waits for the recorded events, accumulates per-kernel elapsed
 * milliseconds and launch counts, and clears the pool.  Kernel ids: */
#define ODTK_KERNEL_PREFILTER 0   /* prefilter_scan_kernel                         */
#define ODTK_KERNEL_SELECT    1   /* select_decode_kernel                          */
#define ODTK_KERNEL_NMS       2   /* nms_kernel                                    */
#define ODTK_KERNEL_IOU       3   /* iou_pairs_kernel                              */
#define ODTK_KERNEL_EPILOGUE  4   /* bias_act_kernel (+ its scalar tail form)      */
#define ODTK_KERNEL_TARGETS   5   /* snap_to_anchors_kernel                        */
#define ODTK_KERNEL_GEMM      6   /* hipBLASLt kernels behind odtk_gemm_bias_act    */
#define ODTK_KERNEL_LOSS      7   /* retina_loss_kernel (forward and backward)      */
#define ODTK_KERNEL_SELHIST   8   /* (rounds 2-3: the selection's histogram launch; no longer launched, id kept) */
#define ODTK_KERNEL_SELFILTER 9   /* (rounds 2-3: the selection's filter launch; no longer launched, id kept)    */
#define ODTK_KERNEL_NMS_ORDER 10  /* nms_kernel, stage 1 (rotated: the first round in order)   */
#define ODTK_KERNEL_NMS_MATRIX 11 /* rotated_sup_matrix_kernel (rotated: pairwise suppression) */
#define ODTK_KERNEL_POOL      12  /* bias_act_maxpool_kernel (the stem's bias + ReLU + max-pool pass)  */
#define ODTK_KERNEL_UPSAMPLE  13  /* upsample_nearest2x_kernel                                         */
#define ODTK_KERNEL_STEM_PACK 14  /* stem_pack_kernel (space-to-depth pack of the network input)        */
#define ODTK_KERNEL_LOSS_REDUCE 15 /* loss_reduce_kernel (second launch of the forward through a workspace)   */
#define ODTK_KERNEL_COUNT     16
/* on: 0 = off, otherwise a bit mask of kernel ids (1 << ODTK_KERNEL_*), -1 = all.  An event pair
 * is a queue marker before and after the kernel: cheap for the 3 post-processing launches of a step,
 * measurably NOT free for the ~110 epilogue launches (about 10 % of an 8.7 ms step), so time those
 * only in a separate pass. */
int odtk_profile_enable(int on);
/* Debug: device buffer (>= 128 KiB, zero-filled) that select_decode / nms workgroups stamp with wall_clock64()
 * (100 MHz) at their phase boundaries -- coarse stamps in the first 64 KiB, select_decode's per-segment fine-phase stamps
 * (16 words per segment, up to 512 segments) in the second; NULL (default) disables.  Not for production use. */
int odtk_debug_set_trace(void *device_buffer);
/* Debug / tuning: launch shape of the loss kernels for one form (which = 0: forward with atomics, 1: backward, 2: forward
 * through a workspace) and head width (fp32_heads = 0: bf16 / fp16, 1: fp32): workgroup size (multiple of 64, <= 1024),
 * resident workgroups per CU the logit walk is capped at (1..64), 16-byte vectors per lane per trip (1, 2 or 4),
 * workgroups of the box-delta walk (1..16384); both workgroup caps are per pyramid level.  The defaults are the measured
 * best; results do not depend on the shape beyond the order of the partial sums.  Process-wide, not
 * thread-safe against concurrent loss launches; a workspace size queried before a change is stale after it. */
int odtk_debug_loss_tuning(int which, int fp32_heads, int threads, int blocks_per_cu, int unroll, int box_blocks);
/* Debug / A-B: memory layout of the loss kernels' logit walk, per form and head width as above.  per_wave (workspace form
 * only, which = 2): 1 = every wave writes its own three sums to the workspace (3 doubles per WAVE of the launch -- the size
 * query follows) and no workgroup barrier closes the walk; window: 0 = the vectors a lane loads per trip lie one grid
 * stride apart, 1 = a wave's vectors of a trip are contiguous in memory; box_rows (read by the backward on
 * channels_last heads): 1 (default) = the box-delta walk enumerates the cells in the order d(deltas) lies in memory and
 * writes one vector per cell, 0 = in depth's order, one element per store (rounds 2-5).  Results do not depend on any of
 * them beyond the order of the partial sums. */
int odtk_debug_loss_layout(int which, int fp32_heads, int per_wave, int window, int box_rows);
/* Debug / A-B: arithmetic form of the classification walk of the loss kernels when gamma == 2 (csrc/loss.hpp).
 * 0: every logit through the symmetric form (one select on the target, one on the sign);  1: 16-byte vectors that hold no
 * positive element and no logit above 64 (all but ~1 in 1000) through the negatives-only form u = exp(x), q = u / (1 + u),
 * ce = ln(1 + u) -- the same three hardware transcendentals, about a third fewer full-rate operations; every other vector
 * and every other gamma as with 0.  Results agree to ~1e-8 (sums) / ~1e-6 of the largest gradient.  The default is
 * ODTK_LOSS_FORM_DEFAULT; process-wide, read once per launch.  2, 3, 4, 6, 7: TIMING ABLATIONS of form 1 for the fp32
 * forward (no depth gather / no arithmetic / no index arithmetic and no depth gather / no box-delta walk / no logit walk;
 * other launches as form 1) -- their sums are wrong on purpose, tools/loss_form_probe.py is their only user, and they
 * exist only in a library built with -DODTK_LOSS_ABLATIONS (`make ablations` -> build_ablate/): the shipped library
 * accepts 0 and 1.  Returns ODTK_ERR_INVALID for any other value. */
#define ODTK_LOSS_FORM_DEFAULT 1
int odtk_debug_loss_form(int form);
int odtk_profile_collect(double total_ms[ODTK_KERNEL_COUNT], int launches[ODTK_KERNEL_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* ODTK_HIP_H */

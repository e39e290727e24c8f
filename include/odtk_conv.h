/*
 * odtk_conv.h -- C ABI of libodtk_conv.so (MI355X / gfx950): the engine-side convolution with a fused epilogue.
 *
 * NOT part of the post-processing drop-in boundary (include/odtk_hip.h) and not needed by it: this library exists so that the
 * inference engine (odtk/fused.py) can drop the separate bias + ReLU pass behind its k x k convolutions.  No reference kernel
 * equivalent: the reference runs conv -> (bias | frozen batch-norm) -> ReLU as separate PyTorch kernels
 * (odtk/backbones/layers.py:5-16, odtk/model.py:57-62, torchvision Bottleneck.forward).
 *
 * Same conventions as odtk_hip.h: extern "C", plain pointers and sizes, device pointers, negative error codes (ODTK_ERR_*),
 * work is only ENQUEUED on the caller's stream -- except the first call for a problem (shape tuple), which times the
 * library's instances on that stream and synchronises with it once per candidate.  Inside a stream capture nothing is timed:
 * the first instance that supports the problem is taken and the problem is re-planned at the next eager call.
 */
#ifndef ODTK_CONV_H
#define ODTK_CONV_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * odtk_conv_bias_act -- y = act( conv2d(x, w) + bias[k] ),  act = ReLU if relu != 0; groups = 1, dilation = 1.
 *   x     [batch, height, width, c_in]                 channels_last (NHWC) activation, `dtype` ODTK_BF16 or ODTK_F16
 *   w     [c_out, kernel_h, kernel_w, c_in]            = a torch weight [c_out, c_in, kh, kw] in channels_last memory format
 *   bias  [c_out]  of `dtype`                          (the instance lists take the bias in the activation type)
 *   y     [batch, out_h, out_w, c_out],  out = (in + 2 pad - kernel) / stride + 1
 * The contraction is a composable_kernel implicit-GEMM convolution on the matrix cores -- the kernels MIOpen itself picks for
 * these layers -- instantiated with an add + clamp epilogue (libdevice_conv_operations.a of the ROCm installation:
 * add_device_grouped_conv2d_fwd_bias_clamp_xdl_nhwgc_gkyxc_nhwgk_{bf16,f16}_*_instances); fp32 accumulation; the instances pass the
 * accumulator through LDS in `dtype` before the epilogue, so the result carries two roundings (acc -> dtype, act(. + bias) ->
 * dtype): exactly what a convolution followed by odtk_bias_act makes, within one ulp of the fp32 result.  Returns ODTK_ERR_UNSUPPORTED when no instance takes the problem (c_in / c_out not a multiple of
 * the instances' vector widths, ...): the caller then runs its convolution + odtk_bias_act pair.
 */
int odtk_conv_bias_act(void *y, const void *x, const void *w, const void *bias, int batch_size, int c_in, int height, int width,
                       int c_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int dtype,
                       int relu, void *stream);

/* ... with different padding before (pad_h, pad_w: top / left) and after (pad_h_end, pad_w_end: bottom / right) the image:
 * out = (in + pad + pad_end - kernel) / stride + 1.  (The space-to-depth form of the ResNet stem is a 4x4 convolution padded (2, 1).) */
int odtk_conv_bias_act_pads(void *y, const void *x, const void *w, const void *bias, int batch_size, int c_in, int height, int width,
                            int c_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w, int pad_h_end,
                            int pad_w_end, int dtype, int relu, void *stream);

/* "#index time-when-chosen instance-name" of the instance the last odtk_conv_bias_act call of this thread ran. */
const char *odtk_conv_last_plan(void);

/*
 * Reproducible plans.  The instance a problem runs on is picked by a stopwatch at its first call, and two instances accumulate in
 * different orders: the same weights then give different bits on different boxes.  odtk_conv_plan_export writes one line per timed
 * problem -- "conv dtype n c h w k r s u v pad_h pad_w pad_h_end pad_w_end index instance-name" -- and returns the bytes the text
 * needs (NUL included; at most `capacity` are written); odtk_conv_plan_import takes such lines (others are ignored) and returns
 * how many it took: the problem then runs on that instance without any timing.  A line whose instance name is not what this
 * build's list has at that index is NOT taken (another ROCm release: the problem is timed as usual).  The engine's plan file
 * carries these lines (odtk/fused.py: plan_state / load_plan, ODTK_CONV_PLAN).
 */
size_t odtk_conv_plan_export(char *text, size_t capacity);
int odtk_conv_plan_import(const char *text);

/*
 * Non-finite accumulators.  The epilogue is composable_kernel's AddClamp, `a > floor ? (a < ceil ? a : ceil) : floor`: a NaN
 * accumulator comes out as `floor` -- 0 on a ReLU layer, -FLT_MAX (-inf after the rounding to 16 bits) on a linear one -- where
 * MIOpen + odtk_bias_act and the reference's PyTorch graph hand the NaN on.  A diverged checkpoint therefore shows up as
 * "no detections", not as NaN scores, on library-routed layers; `ODTK_CHECK_FINITE=1` makes the engine assert that its head
 * tensors are finite (odtk/fused.py).
 */

/* Instances linked for `dtype` (237 bf16, 228 fp16 with ROCm 7.2's archive); 0 for any other dtype. */
int odtk_conv_instance_count(int dtype);

#ifdef __cplusplus
}
#endif
#endif

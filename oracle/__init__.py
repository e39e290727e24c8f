"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's per-anchor post-processing path
(NVIDIA/retinanet-examples: odtk/box.py decode/nms, csrc/cuda/decode_rotate.cu,
csrc/cuda/nms_iou.cu).  Nothing under this directory is part of the product:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(retinanet-examples_amd/) never imports oracle/ and has no CPU fallback.

Pinning status
--------------
* axis-aligned decode / nms / generate_anchors: PINNED -- tests/golden/*.npz were
  produced by running the reference's own odtk/box.py (imported from
  /root/reference through oracle/ref_loader.py) in the build container; the
  restatement in box_oracle.py reproduces them bit-for-bit
  (tests/test_oracle_golden.py).  Anchors are additionally pinned to the rounded
  known-answer table embedded in the reference at extras/cppapi/export.cpp:69-75.
* rotated IoU / rotated NMS: PINNED to the reference's own source.  The native library as a whole
  cannot be built here (nvcc / thrust / cub / TensorRT absent), but the device code that does the
  geometry -- Vector / Line / IntersectionArea, nms_rotate_kernel, iou_cuda_kernel
  (csrc/cuda/nms_iou.cu:41-258, :324-375) -- is plain C++ once a dozen CUDA names exist:
  oracle/ref_build/build_ref.py reads those lines from /root/reference where they lie, wraps them
  between prelude.hpp and harness.cpp and builds oracle/_ref/libodtk_ref_native.so with g++ in IEEE
  mode (the reference's nvcc build used --use_fast_math, whose bits nothing else reproduces).
  tests/golden/rotated_ref_*.npz hold its outputs (oracle/gen_golden_native.py); the C restatement
  oracle/c/odtk_oracle.c reproduces them bit for bit, live runs included
  (tests/test_oracle_native_ref.py), and so do the HIP kernels (tests/test_gpu_rotated.py).
* rotated (and axis-aligned) decode, CUDA path: the per-detection gather + box lambdas of
  csrc/cuda/decode.cu:121-159 and decode_rotate.cu:116-167 are compiled the same way and applied to the
  indices the CPU-convention selection keeps (tests/golden/decode_ref_*.npz): index decomposition, delta
  gather layout, classes, scores and the sin/cos passthrough are reproduced exactly; the box agrees
  within 1e-4 once the CPU path's two-sided clamp is applied (the CUDA lambda clamps one side, sums the
  centre in another order and uses a float exp -- documented conventions, the CPU path is normative).
  Threshold / top-k selection of the rotated decode has no runnable reference of its own; it is the
  same code as the axis-aligned selection, which is pinned to box.py.
"""

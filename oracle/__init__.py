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
* rotated decode / rotated IoU / rotated NMS: PARITY UNPINNED.  The reference has
  no runnable CPU implementation (odtk/box.py:408 NameError, box.py:303 shape bug)
  and its CUDA sources cannot be built here (nvcc/thrust/cub/TensorRT absent), so
  oracle/c/odtk_oracle.c restates csrc/cuda/nms_iou.cu + decode_rotate.cu and is
  cross-checked only against an independent float64 convex-polygon clipper.
"""

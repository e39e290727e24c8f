#!/bin/bash
# Round 6, GPU call 8: NMS -- the push over everything gets eight boxes, then sorted rounds of batched pushes take over.
O=gpurun_out/r6c8; mkdir -p $O
export TMPDIR=/tmp
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/nms_clustered_chunks$C.txt | head -19 | cut -c1-300; grep "launch, event" $O/nms_clustered_chunks$C.txt
done
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_clustered_probe.py --first 10 ) > $O/nms_clustered_first10.txt 2>&1; grep "img 0 phases" $O/nms_clustered_first10.txt | cut -c1-900
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_default.txt 2>&1; grep "back to back, event" $O/nms_rn101_default.txt | cut -c1-600
( ODTK_NMS_CHUNKS=1 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_chunks.txt 2>&1; grep "back to back, event" $O/nms_rn101_chunks.txt | cut -c1-600
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py tests/test_gpu_threads.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt

#!/bin/bash
# Round 6, GPU call 6: NMS -- first round through the chunk loop, later rounds through the batched push, no "push over
# everything" (ODTK_NMS_CHUNKS=1: rounds 3-5's form) on the trained detector's candidates, on the bench, on RN101 bs 16.
O=gpurun_out/r6c6; mkdir -p $O
export TMPDIR=/tmp
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/nms_clustered_probe.py --first 10 ) > $O/nms_clustered_first10_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/nms_clustered_first10_chunks$C.txt | head -24 | cut -c1-400
done
for C in 0 1 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_chunks${C}_detail.json ) > $O/bench_chunks$C.json 2> $O/bench_chunks$C.err
  python - <<P
import json
d = json.loads(open('$O/bench_chunks$C.json').read().strip().splitlines()[-1])
print('ODTK_NMS_CHUNKS=$C', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
done
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_batched.txt 2>&1; grep "back to back, event\|img  0\|img 0 phases" $O/nms_rn101_batched.txt | cut -c1-600
( ODTK_NMS_CHUNKS=1 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_chunks.txt 2>&1; grep "back to back, event\|img  0\|img 0 phases" $O/nms_rn101_chunks.txt | cut -c1-600
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt

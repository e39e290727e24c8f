#!/bin/bash
# Round 6, GPU call 4: the NMS's batched push (default) against the chunk loop (ODTK_NMS_CHUNKS=1) on a trained detector's
# candidates and on the bench; parity suites.
O=gpurun_out/r6c4; mkdir -p $O
export TMPDIR=/tmp
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/nms_clustered_chunks$C.txt | head -30
done
for C in 0 1 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_chunks${C}_detail.json ) > $O/bench_chunks$C.json 2> $O/bench_chunks$C.err
  python - <<P
import json
d = json.loads(open('$O/bench_chunks$C.json').read().strip().splitlines()[-1])
print('ODTK_NMS_CHUNKS=$C', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
done
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/trace_postproc.py ) > $O/trace_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/trace_chunks$C.txt | grep -A9 "nms phases"
done
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_batched.txt 2>&1; grep "back to back, event" $O/nms_rn101_batched.txt
( ODTK_NMS_CHUNKS=1 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_chunks.txt 2>&1; grep "back to back, event" $O/nms_rn101_chunks.txt
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py tests/test_gpu_threads.py tests/test_gpu_graph.py tests/test_compiled_binding.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -5 $O/pytest_parity.txt

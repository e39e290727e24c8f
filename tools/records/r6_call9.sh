#!/bin/bash
# Round 6, GPU call 9: the NMS's class-parallel resolve (default) against round 6's batched push (ODTK_NMS_CLASSES=0): a trained
# detector's candidates, RN101 bs 16 heads, the bench's heads; parity suites; a short bench.
O=gpurun_out/r6c9; mkdir -p $O
export TMPDIR=/tmp
for C in 1 0; do
  ( ODTK_NMS_CLASSES=$C timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered_classes$C.txt 2>&1; echo "== ODTK_NMS_CLASSES=$C"; grep -v amdgpu.ids $O/nms_clustered_classes$C.txt | head -20 | cut -c1-400; grep "launch, event" $O/nms_clustered_classes$C.txt
  ( ODTK_NMS_CLASSES=$C timeout 300 python tools/nms_clustered_probe.py --generic ) > $O/nms_clustered_generic_classes$C.txt 2>&1; grep "bit for bit\|launch, event" $O/nms_clustered_generic_classes$C.txt
  ( ODTK_NMS_CLASSES=$C timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_classes$C.txt 2>&1; grep "back to back, event\|img 0 phases" $O/nms_rn101_classes$C.txt | cut -c1-700
  ( ODTK_NMS_CLASSES=$C timeout 300 python tools/nms_trace_probe.py ) > $O/nms_rn50_classes$C.txt 2>&1; grep "back to back, event\|img 0 phases" $O/nms_rn50_classes$C.txt | cut -c1-700
done
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py tests/test_gpu_threads.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt
for C in 1 0; do
( ODTK_NMS_CLASSES=$C timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_detail_classes$C.json ) > $O/bench_classes$C.json 2> $O/bench_classes$C.err
python - <<P
import json
d = json.loads(open('$O/bench_classes$C.json').read().strip().splitlines()[-1])
print('bench classes=$C', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
done

#!/bin/bash
# Round 6, GPU call 7: NMS with the adaptive choice between the push over everything and sorted rounds of batched pushes:
# trained detector's candidates, RN101 bs 16 heads, the bench; then the whole GPU suite on the round's defaults.
O=gpurun_out/r6c7; mkdir -p $O
export TMPDIR=/tmp
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/nms_clustered_chunks$C.txt | head -19 | cut -c1-300; grep "launch, event" $O/nms_clustered_chunks$C.txt
done
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_default.txt 2>&1; grep "back to back, event" $O/nms_rn101_default.txt | cut -c1-600
( ODTK_NMS_CHUNKS=1 timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101_chunks.txt 2>&1; grep "back to back, event" $O/nms_rn101_chunks.txt | cut -c1-600
( timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err
python - <<P
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt

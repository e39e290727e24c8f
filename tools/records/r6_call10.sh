#!/bin/bash
# Round 6, GPU call 10: the axis-aligned NMS resolves a chunk / a batch from suppression COLUMNS by a fixpoint (resolve_columns)
# instead of walking the candidates: a trained detector's candidates, RN101 bs 16 heads, the bench's heads; the whole GPU suite; a short bench.
O=gpurun_out/r6c10; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered.txt 2>&1; grep -v amdgpu.ids $O/nms_clustered.txt | head -24 | cut -c1-400; grep "launch, event" $O/nms_clustered.txt
( timeout 300 python tools/nms_clustered_probe.py --generic ) > $O/nms_clustered_generic.txt 2>&1; grep "bit for bit\|launch, event" $O/nms_clustered_generic.txt
( timeout 300 python tools/nms_trace_probe.py --backbone ResNet101FPN --batch 16 ) > $O/nms_rn101.txt 2>&1; grep "back to back, event\|img 0 phases" $O/nms_rn101.txt | cut -c1-700
( timeout 300 python tools/nms_trace_probe.py ) > $O/nms_rn50.txt 2>&1; grep "back to back, event\|img 0 phases" $O/nms_rn50.txt | cut -c1-700
( timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err
python - <<P
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'), d.get('latency_bound'))
P
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt

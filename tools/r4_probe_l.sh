#!/bin/bash
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_model.py tests/test_gpu_graph.py tests/test_gpu_configs.py -x -q -m gpu > $O/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a $O/summary.txt
tail -6 $O/tests1.log
timeout 500 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4l/bench.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['kernels'])); print(json.dumps(d['epilogue_roofline'])); print(json.dumps(d['marker_timed'])); print(json.dumps(d['latency_bound']))
PY
timeout 600 python tools/bf16_ablation.py > $O/bf16_ablation.txt 2> $O/bf16_ablation.err; tail -14 $O/bf16_ablation.txt; tail -3 $O/bf16_ablation.err

#!/bin/bash
# Round 5, GPU call 2: the convolution library (tests, the engine's per-layer plan, step time with / without it), the rest of the
# GPU suite after the stop of call 1, the rotated training probe at bench.py's own configuration.
O=gpurun_out/r5c2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_conv_library.py -x -q ) > $O/pytest_conv.txt 2>&1; tail -15 $O/pytest_conv.txt
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_conv_plan_detail.json ) > $O/bench_conv_plan.json 2> $O/bench_conv_plan.err; tail -3 $O/bench_conv_plan.err; tail -1 $O/bench_conv_plan.json | cut -c1-600
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --no-conv-library --detail-out $O/bench_no_conv_detail.json ) > $O/bench_no_conv.json 2> $O/bench_no_conv.err; tail -1 $O/bench_no_conv.json | cut -c1-400
python - <<'P'
import json
d = json.load(open('gpurun_out/r5c2/bench_conv_plan_detail.json'))
ce = d.get('conv_epilogue') or {}
print({k: v for k, v in ce.items() if k != 'layers'})
for k, v in (ce.get('layers') or {}).items(): print(k, v)
P
( time timeout 600 python -m pytest tests -m gpu -q --deselect tests/test_gpu_conv_library.py ) > $O/pytest_gpu_all.txt 2>&1; tail -8 $O/pytest_gpu_all.txt
( time timeout 300 python tools/rotated_train_probe.py --bench-like --steps 40 ) > $O/rotated_train_bench_like.txt 2>&1; grep -n "non-finite\|==" $O/rotated_train_bench_like.txt

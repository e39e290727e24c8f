#!/bin/bash
# Round 5, GPU call 8: the stem in space-to-depth form (tests, plan A/B, step time), engine tests around it
O=gpurun_out/r5c8; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_conv_library.py -q ) > $O/pytest_conv.txt 2>&1; tail -5 $O/pytest_conv.txt
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('conv_epilogue'))"
python - <<'P'
import json
d = json.load(open('gpurun_out/r5c8/bench_detail.json'))
for k, v in ((d.get('conv_epilogue') or {}).get('layers') or {}).items():
    if 'stem' in k: print(k, v)
P
tail -3 $O/bench.err
( time timeout 900 python -m pytest tests/test_gpu_fused_model.py tests/test_gpu_detection_parity.py tests/test_gpu_graph.py tests/test_gpu_cli.py tests/test_gpu_fused.py -q ) > $O/pytest_engine.txt 2>&1; tail -6 $O/pytest_engine.txt

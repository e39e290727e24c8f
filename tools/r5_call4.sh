#!/bin/bash
# Round 5, GPU call 4: select_decode (gather under the merge levels, the finisher's one round trip) -- parity first, then time
O=gpurun_out/r5c4; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_conv_library.py -q ) > $O/pytest_conv.txt 2>&1; tail -4 $O/pytest_conv.txt
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_rotated.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_graph.py -q ) > $O/pytest_select.txt 2>&1; tail -6 $O/pytest_select.txt
for i in 1 2; do python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernels_us'])"; done
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_avg_us'], d.get('conv_epilogue'))"
python - <<'P'
import json
d = json.load(open('gpurun_out/r5c4/bench_detail.json'))
for k, v in ((d.get('conv_epilogue') or {}).get('layers') or {}).items():
    if 'only' in k or not v['library']: print(k, v)
P
python tools/trace_postproc.py > $O/trace_postproc.txt 2>&1; tail -30 $O/trace_postproc.txt

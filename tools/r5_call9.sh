#!/bin/bash
# Round 5, GPU call 9: strided 1x1 routing, the whole GPU suite, smoke, RCCL initialisation
O=gpurun_out/r5c9; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('conv_epilogue'), d['kernels_avg_us'])"
python - <<'P'
import json
d = json.load(open('gpurun_out/r5c9/bench_detail.json'))
for k, v in ((d.get('conv_epilogue') or {}).get('layers') or {}).items():
    if 'down' in k or 'stem' in k or not v['library']: print(k, v)
P
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu_all.txt 2>&1; tail -6 $O/pytest_gpu_all.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
( time timeout 200 python tools/rccl_init_probe.py ) > $O/rccl_init_probe.txt 2>&1; tail -9 $O/rccl_init_probe.txt

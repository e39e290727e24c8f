#!/bin/bash
# Round 5, GPU call 11: the driver's own command, once more on the final tree
O=gpurun_out/r5c11; mkdir -p $O
export TMPDIR=/tmp
( time timeout 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r5c11/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'latency_bound', 'kernels_avg_us', 'postproc_us_per_step', 'conv_epilogue', 'conv_roofline', 'quoted', 'other_configs')}, indent=1)[:3500])
print('cpu_baseline', d.get('cpu_baseline'))
P
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null
tail -3 $O/bench_driver_cmd.err

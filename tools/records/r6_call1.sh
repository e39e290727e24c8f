#!/bin/bash
# Round 6, GPU call 1: the new tests (plan replay, trained-detector AP), the long trained-AP run on two training seeds, the
# driver's bench command with every BASELINE configuration inside its default budget, then the whole GPU suite.
O=gpurun_out/r6c1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_conv_plan.py tests/test_gpu_trained_ap.py -m gpu -x -q -s ) > $O/new_tests.txt 2>&1; tail -12 $O/new_tests.txt
( time timeout 1500 python tools/trained_ap.py --seeds 0 1 --iterations 2500 --images 256 --json $O/trained_ap.json ) > $O/trained_ap.txt 2> $O/trained_ap.err; cat $O/trained_ap.txt; tail -3 $O/trained_ap.err
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r6c1/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'latency_bound', 'kernels_avg_us', 'postproc_us_per_step', 'conv_epilogue', 'parity', 'other_configs', 'dropped_to_fit')}, indent=None)[:4000])
P
tail -4 $O/bench_driver_cmd.err
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt

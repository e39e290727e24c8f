#!/bin/bash
# Round 6, GPU call 21: the whole GPU suite, smoke() and the driver's bench command on the tree with the changed bindings.
O=gpurun_out/r6c21; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; grep -v amdgpu $O/pytest_gpu.txt | tail -4
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r6c21/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'kernels_avg_us', 'postproc_us_per_step', 'parity')}, indent=None)[:1500])
print({k: v.get('value') for k, v in d['other_configs'].items()})
P

#!/bin/bash
# Round 5, GPU call 1: the whole GPU suite, the driver's bench command (the line must parse), bare --gpus 1,
# the rotated training probe, the loss PMC kit.  Everything under gpurun_out/r5c1/.
O=gpurun_out/r5c1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
( time timeout 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r5c1/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('bench line ok:', len(s[-1]), 'bytes; value', d.get('value'), d.get('unit'), 'roofline', d.get('roofline'), 'cpu', d.get('cpu_baseline'))
P
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null
tail -4 $O/bench_driver_cmd.err
( time timeout 300 python tools/rotated_train_probe.py --steps 60 ) > $O/rotated_train_probe.txt 2>&1
tail -12 $O/rotated_train_probe.txt
timeout 500 bash tools/loss_pmc.sh > $O/loss_pmc.txt 2>&1
tail -40 $O/loss_pmc.txt

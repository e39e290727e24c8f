#!/bin/bash
# Round 5, GPU call 15: the round's profiles and the driver's bench command on the FINAL kernels (cooperative selection route)
O=gpurun_out/r5c15; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r05 ) > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
grep -n "prefilter_scan\|select_decode\|nms_kernel\|steady window" gpurun_out/prof_r05/r05_bench_steady_kernel_stats.txt | cut -c1-150
cat gpurun_out/prof_r05/r05_pmc_traffic.json | head -12
tail -2 gpurun_out/prof_r05/r05_pmc_mfma_bench.txt
( time timeout 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r5c15/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'latency_bound', 'kernels_avg_us', 'postproc_us_per_step', 'conv_epilogue', 'other_configs')}, indent=None)[:3000])
P
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null
( time timeout 120 python tools/trace_postproc.py ) > $O/trace_postproc.txt 2>&1; head -14 $O/trace_postproc.txt

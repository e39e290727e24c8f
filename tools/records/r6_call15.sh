#!/bin/bash
# Round 6, GPU call 15 (= call 11 again, after the last change to the kernel sources: the partition knobs): the round's profiles (tools/profile_round.sh r06: rocprofv3 kernel stats of the bench command, PMC traffic of the
# prefilter, MFMA utilisation) and the driver's bench command on the round's final kernel sources.
O=gpurun_out/r6c15; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r06 ) > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
grep -n "prefilter_scan\|select_decode\|nms_kernel\|steady window" gpurun_out/prof_r06/r06_bench_steady_kernel_stats.txt | cut -c1-150
cat gpurun_out/prof_r06/r06_pmc_traffic.json | head -12
tail -2 gpurun_out/prof_r06/r06_pmc_mfma_bench.txt
cp gpurun_out/prof_r06/r06_pmc_traffic.json profiles/ 2>/dev/null   # (the bench below then quotes it: same kernel sources)
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r6c15/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'latency_bound', 'kernels_avg_us', 'postproc_us_per_step', 'conv_epilogue', 'parity', 'other_configs')}, indent=None)[:3500])
P
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null
( time timeout 120 python tools/trace_postproc.py ) > $O/trace_postproc.txt 2>&1; head -30 $O/trace_postproc.txt | cut -c1-250

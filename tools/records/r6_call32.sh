#!/bin/bash
# Round 6, GPU call 32 (= call 31 again): the pool pass tests negative NaNs against the key of -inf of ITS dtype (fp16: 0x83ff; call 31 found the bf16 constant used for both)
# -> the round's profiles again on these kernel sources (tools/profile_round.sh r06), the
# whole GPU suite, smoke() and the driver's bench command.
O=gpurun_out/r6c32; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r06 ) > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
grep -n "prefilter_scan\|select_decode\|nms_kernel\|steady window" gpurun_out/prof_r06/r06_bench_steady_kernel_stats.txt | cut -c1-150
cat gpurun_out/prof_r06/r06_pmc_traffic.json | head -12
tail -2 gpurun_out/prof_r06/r06_pmc_mfma_bench.txt
cp gpurun_out/prof_r06/r06_pmc_traffic.json profiles/ 2>/dev/null   # (the bench and the guard test below then see it: same kernel sources)
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; grep -v amdgpu $O/pytest_gpu.txt | tail -4
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python - <<'P'
import json
s = open('gpurun_out/r6c32/bench_driver_cmd.json').read().strip().splitlines()
def bad(x): raise ValueError(x)
d = json.loads(s[-1], parse_constant=bad)
print('line:', len(s[-1]), 'bytes')
print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step', 'roofline', 'kernels_avg_us', 'postproc_us_per_step', 'parity')}, indent=None)[:1500])
print(json.dumps(d['other_configs'])[:2500])
P
cp gpurun_out/bench_detail_latest.json $O/bench_detail_driver_cmd.json 2>/dev/null

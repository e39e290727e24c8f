#!/bin/bash
# Round 6, GPU call 3: NMS on a trained detector's candidates (baseline trace), rank-by-counting with wide bins (A/B), the
# cooperative route's behaviour inside the step (default timeout and 5 us), parity suites.
O=gpurun_out/r6c3; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/nms_clustered_probe.py ) > $O/nms_clustered_sorted_runs.txt 2>&1; grep -v amdgpu.ids $O/nms_clustered_sorted_runs.txt | head -60
( timeout 300 python tools/nms_clustered_probe.py --generic ) > $O/nms_clustered_generic.txt 2>&1; grep -v amdgpu.ids $O/nms_clustered_generic.txt | head -24
for R in 1 0; do
  ( ODTK_SELECT_RANK=$R timeout 300 python tools/trace_postproc.py ) > $O/trace_rank$R.txt 2>&1; echo "== ODTK_SELECT_RANK=$R"; grep -v amdgpu.ids $O/trace_rank$R.txt | head -7
done
for R in 1 0 1 0; do
  ( ODTK_SELECT_RANK=$R timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_rank${R}_detail.json ) > $O/bench_rank$R.json 2> $O/bench_rank$R.err
  python - <<P
import json
d = json.loads(open('$O/bench_rank$R.json').read().strip().splitlines()[-1])
print('ODTK_SELECT_RANK=$R', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
done
( timeout 600 python tools/select_routes_instep.py --steps 200 ) > $O/select_routes_default.txt 2>&1; grep -v amdgpu.ids $O/select_routes_default.txt
( ODTK_SELECT_COOP_TICKS=500 timeout 600 python tools/select_routes_instep.py --steps 200 ) > $O/select_routes_500.txt 2>&1; grep -v amdgpu.ids $O/select_routes_500.txt
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py tests/test_gpu_select_routes.py tests/test_gpu_rotated.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -5 $O/pytest_parity.txt

#!/bin/bash
# Round 6, GPU call 2: select_decode orders by counting + decodes in place (ODTK_SELECT_RANK A/B), parity suites on it, the whole
# GPU suite, a trained model's NMS inputs saved for offline study.
O=gpurun_out/r6c2; mkdir -p $O
export TMPDIR=/tmp
for R in 1 0; do
  ( ODTK_SELECT_RANK=$R timeout 300 python tools/trace_postproc.py ) > $O/trace_rank$R.txt 2>&1; echo "== ODTK_SELECT_RANK=$R"; grep -v amdgpu.ids $O/trace_rank$R.txt | head -14
done
for R in 1 0 1 0; do
  ( ODTK_SELECT_RANK=$R timeout 400 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_rank${R}_detail.json ) > $O/bench_rank$R.json 2> $O/bench_rank$R.err
  python - <<P
import json
d = json.loads(open('$O/bench_rank$R.json').read().strip().splitlines()[-1])
print('ODTK_SELECT_RANK=$R', d['value'], d['ms_per_step'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'))
P
done
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_nms_corners.py tests/test_gpu_select_routes.py tests/test_gpu_rotated.py tests/test_gpu_threads.py -q -x ) > $O/pytest_parity.txt 2>&1; tail -8 $O/pytest_parity.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt
( time timeout 600 python tools/trained_ap.py --seeds 0 --iterations 1500 --images 64 --save-postproc-inputs $O/trained_postproc_inputs.npz ) > $O/trained_ap_short.txt 2> $O/trained_ap_short.err; tail -8 $O/trained_ap_short.txt; tail -2 $O/trained_ap_short.err

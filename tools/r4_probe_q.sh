#!/bin/bash
# round 4, probe q: rank-merge sorts whose work follows n_valid (no padding merged)
O=gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fused.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
COMMON="--dtype bf16 --logits --channels-last --kind sparse --batch 8 --iters 200 --bias --table"
timeout 200 python tools/postproc_bench.py $COMMON > $O/pp_table.json 2> $O/pp_table.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4q/pp_table.json')); print(d['kernels_us_per_call'])
PY

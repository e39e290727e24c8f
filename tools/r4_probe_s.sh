#!/bin/bash
# round 4, probe s: box gather overlapped with the ranking (axis-aligned probe rounds, rotated stage 1)
O=gpurun_out/r4s; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_rotated.py tests/test_gpu_nms_corners.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
COMMON="--dtype bf16 --logits --channels-last --kind sparse --batch 8 --iters 200 --bias --table"
timeout 100 python tools/postproc_bench.py $COMMON > $O/pp_axis.json 2> $O/pp_axis.err
timeout 100 python tools/postproc_bench.py $COMMON --rotated --anchors 27 --iters 50 > $O/pp_rot.json 2> $O/pp_rot.err
python - <<'PY'
import json
for f in ('axis', 'rot'):
    try:
        d = json.load(open('gpurun_out/r4s/pp_%s.json' % f)); print(f, d['kernels_us'], d['kernels_us_per_call'])
    except Exception as e:
        print(f, 'failed', e)
PY

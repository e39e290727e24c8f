#!/bin/bash
# round 4, probe p: rotated NMS rework (stage 1 by repeated probes, 4-row matrix slices, LDS-resident matrix in the resolve)
O=gpurun_out/r4p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_rotated.py tests/test_gpu_nms_corners.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
COMMON="--no-other-configs --no-eager-leg --cpu-seconds 0 --steps 20 --warmup 5"
timeout 300 python bench.py $COMMON --rotated-bbox > $O/bench_rot.json 2> $O/bench_rot.err; echo "rot rc=$?"
timeout 300 python bench.py $COMMON --rotated-bbox --unit-rotation > $O/bench_rot_unit.json 2> $O/bench_rot_unit.err; echo "unit rc=$?"
timeout 300 python bench.py $COMMON --backbone ResNet101FPN --batch 16 > $O/bench_rn101.json 2> $O/bench_rn101.err; echo "rn101 rc=$?"
python - <<'PY'
import json
for f in ('rot', 'rot_unit', 'rn101'):
    try:
        d = json.loads(open('gpurun_out/r4p/bench_%s.json' % f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('kernels'), d.get('latency_bound', {}).get('nms_kernel'))
    except Exception as e:
        print(f, 'failed', e)
PY

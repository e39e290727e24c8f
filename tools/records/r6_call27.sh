#!/bin/bash
# Round 6, GPU call 27: the stem's pool pass on packed order-preserving keys with unconditional (clamped) loads, stem_pack with
# float2 loads -- tests that cover them, then tools/pool_probe.py and the per-kernel times of a short bench run.
O=gpurun_out/r6c27; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_conv_library.py tests/test_conv_plan.py -m gpu -q -x ) > $O/pytest.txt 2>&1; grep -v amdgpu $O/pytest.txt | tail -5
( timeout 200 python tools/pool_probe.py ) > $O/pool_probe.txt 2>&1; grep -v amdgpu $O/pool_probe.txt | tail -4
( time timeout 600 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs ) > $O/bench_short.json 2> $O/bench_short.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r6c27/bench_short.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('kernels_avg_us'), d.get('parity', {}).get('box_coords_beyond_1e-4'))
print(json.dumps(d.get('epilogue_roofline'))[:800])
P
grep -n "maxpool\|stem_pack\|upsample" gpurun_out/bench_detail_latest.json | head -20 | cut -c1-200

#!/bin/bash
# Round 5, GPU call 13: the cooperative route, lighter form (global threshold, keys >= T published unsorted, one-trip load, standard sort)
O=gpurun_out/r5c13; mkdir -p $O
export TMPDIR=/tmp
( time timeout 120 python tools/trace_postproc.py ) > $O/trace_postproc.txt 2>&1; head -14 $O/trace_postproc.txt
for i in 1 2; do timeout 120 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coop   ', d['wall_us_per_call'], d['kernels_us'])"; done
ODTK_SELECT_COOP_TICKS=0 timeout 120 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('off    ', d['wall_us_per_call'], d['kernels_us'])"
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x ) > $O/pytest_coop.txt 2>&1; tail -4 $O/pytest_coop.txt
for v in 3000 0; do ( ODTK_SELECT_COOP_TICKS=$v timeout 200 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs ) 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ticks=$v', d['value'], d['kernels_avg_us'])"; done

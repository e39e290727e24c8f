#!/bin/bash
# Round 5, GPU call 12: select_decode's cooperative route (segment-local barrier, global threshold, ranked merge of sorted runs)
O=gpurun_out/r5c12; mkdir -p $O
export TMPDIR=/tmp
SUITES="tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_rotated.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_graph.py tests/test_gpu_threads.py tests/test_gpu_nms_corners.py"
( time timeout 120 python tools/trace_postproc.py ) > $O/trace_postproc.txt 2>&1; head -16 $O/trace_postproc.txt
for i in 1 2; do timeout 120 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coop   ', d['wall_us_per_call'], d['kernels_us'])"; done
ODTK_SELECT_COOP_TICKS=0 timeout 120 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('off    ', d['wall_us_per_call'], d['kernels_us'])"
ODTK_SELECT_COOP_TICKS=1 timeout 120 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ticks=1', d['wall_us_per_call'], d['kernels_us'])"
( time timeout 900 python -m pytest $SUITES -q -x ) > $O/pytest_coop.txt 2>&1; tail -5 $O/pytest_coop.txt
( time ODTK_SELECT_COOP_TICKS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_threads.py -q -x ) > $O/pytest_ticks1.txt 2>&1; tail -5 $O/pytest_ticks1.txt

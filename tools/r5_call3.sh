#!/bin/bash
# Round 5, GPU call 3: select_decode with the gather in front of the sort -- parity suites that exercise it, then its time
O=gpurun_out/r5c3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests/test_gpu_conv_library.py -q ) > $O/pytest_conv.txt 2>&1; tail -6 $O/pytest_conv.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_rotated.py tests/test_gpu_fused.py tests/test_gpu_configs.py -q -x ) > $O/pytest_select.txt 2>&1; tail -6 $O/pytest_select.txt
python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 > $O/postproc_bench.txt 2>&1; tail -4 $O/postproc_bench.txt
( time timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_detail.json ) > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels_avg_us'], d['latency_bound'])"

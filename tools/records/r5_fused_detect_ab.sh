#!/bin/bash
# Round 5, GPU call 5: odtk_detect as prefilter + ONE selection/decode/NMS launch (csrc/detect.hpp) -- parity, then time, A/B against
# the three-launch form (ODTK_NO_FUSED_NMS=1)
O=gpurun_out/r5c5; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_nms_corners.py tests/test_gpu_rotated.py tests/test_gpu_fused.py tests/test_gpu_configs.py tests/test_gpu_graph.py tests/test_gpu_threads.py tests/test_gpu_detection_parity.py -q -x ) > $O/pytest_detect.txt 2>&1; tail -6 $O/pytest_detect.txt
for v in 0 1; do
  ODTK_NO_FUSED_NMS=$v python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_FUSED=$v', d['wall_us_per_call'], d['kernels_us'])"
done
for v in 0 1; do
  ( ODTK_NO_FUSED_NMS=$v timeout 300 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --detail-out $O/bench_detail_nofused$v.json ) > $O/bench_nofused$v.json 2> $O/bench_nofused$v.err; tail -1 $O/bench_nofused$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NO_FUSED=$v', d['value'], d['kernels_avg_us'], d.get('postproc_us_per_step'))"
done
( timeout 200 python bench.py --steps 20 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs --rotated-bbox --detail-out $O/bench_detail_rot.json ) > $O/bench_rot.json 2> $O/bench_rot.err; tail -1 $O/bench_rot.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rotated', d['value'], d['kernels_avg_us'], d.get('postproc_us_per_step'))"

#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a $O/summary.txt
tail -5 $O/tests1.log
timeout 300 python tools/trace_postproc.py > $O/trace.txt 2>&1
timeout 200 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 > $O/pp_bf16_sparse.json 2> $O/pp_bf16_sparse.err
timeout 200 python tools/postproc_bench.py --kind sparse --batch 8 --iters 20 > $O/pp_fp32_sparse.json 2> $O/pp_fp32_sparse.err
timeout 200 python tools/postproc_bench.py --kind dense --dtype bf16 --logits --channels-last --bias --batch 8 --iters 20 > $O/pp_bf16_dense.json 2> $O/pp_bf16_dense.err
timeout 200 python tools/postproc_bench.py --kind saturated --batch 8 --iters 5 > $O/pp_fp32_saturated.json 2> $O/pp_fp32_saturated.err
timeout 200 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 20 --rotated --anchors 27 > $O/pp_bf16_rot.json 2> $O/pp_bf16_rot.err
timeout 400 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/trace.txt | tail -20
for f in $O/pp_*.json $O/bench*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print({k:d[k] for k in ('wall_us_per_call','kernels_us','prefilter_GBps','value','ms_per_step','kernels','roofline') if k in d})
except Exception as e:
    print('unreadable', e)
PY
done

#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
export TMPDIR=/tmp
for sp in 2 4 1; do
  ODTK_SCAN_SPAN=$sp timeout 200 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 30 > $O/pp_span$sp.json 2> $O/pp_span$sp.err
  python -c "
import json;d=json.loads(open('$O/pp_span$sp.json').read().strip().split('\n')[-1]);print('span $sp alone', d['kernels_us'])"
done
for sp in 2 4; do
  ODTK_SCAN_SPAN=$sp timeout 400 python bench.py --steps 30 --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs > $O/bench_span$sp.json 2> $O/bench_span$sp.err
  python -c "
import json;d=json.loads(open('$O/bench_span$sp.json').read().strip().split('\n')[-1]);print('span $sp step', d['value'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
ODTK_SCAN_SPAN=4 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -3

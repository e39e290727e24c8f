#!/bin/bash
# round 4, probe o: the prefilter's precomputed threshold table: parity, then alone and in step, with and without
O=gpurun_out/r4o; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_model.py -x -q -m gpu 2>&1 | tail -3
COMMON="--dtype bf16 --logits --channels-last --kind sparse --batch 8 --iters 200"
for rep in 1 2; do
  timeout 200 python tools/postproc_bench.py $COMMON --bias > $O/pp_bias_$rep.json 2> $O/pp_bias_$rep.err
  timeout 200 python tools/postproc_bench.py $COMMON --bias --table > $O/pp_table_$rep.json 2> $O/pp_table_$rep.err
done
timeout 200 python tools/postproc_bench.py $COMMON > $O/pp_nobias.json 2> $O/pp_nobias.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r4o/pp_*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], d['kernels_us_per_call'])
PY
ODTK_NO_THRESHOLD_TABLE=1 timeout 400 python bench.py --no-other-configs > $O/bench_notable.json 2> $O/bench_notable.err; tail -c 1500 $O/bench_notable.json
timeout 400 python bench.py --no-other-configs > $O/bench_table.json 2> $O/bench_table.err; tail -c 1500 $O/bench_table.json

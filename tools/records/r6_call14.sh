#!/bin/bash
# Round 6, GPU call 14: (a) the whole GPU suite + smoke() on the round's tree; (b) select_decode's partition swept: workgroups
# provided per segment (ODTK_SELECT_SPANS_PER_PART) x candidates per participating workgroup (ODTK_SELECT_KEYS_PER_PART).
O=gpurun_out/r6c14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for cfg in "44 3072" "22 1536" "11 1536" "11 1024" "22 2048" "8 768" "16 1280"; do
  set -- $cfg
  export ODTK_SELECT_SPANS_PER_PART=$1 ODTK_SELECT_KEYS_PER_PART=$2
  ( timeout 300 python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-other-configs --no-eager-leg --cpu-seconds 0 --detail-out $O/bench_detail_$1_$2.json ) > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  ( timeout 120 python tools/trace_postproc.py ) > $O/trace_$1_$2.txt 2>&1
  python - <<P
import json
d = json.loads(open('$O/bench_$1_$2.json').read().strip().splitlines()[-1])
print('spans/part $1 keys/part $2:', d['value'], d.get('kernels_avg_us'), d.get('postproc_us_per_step'), d.get('parity', {}).get('scores_bit_exact'))
P
  grep "^  P[345]:" $O/trace_$1_$2.txt | head -6 | cut -c1-120
done

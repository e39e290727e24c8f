#!/bin/bash
# the prefilter's in-step penalty is address translation (profiles/r05_instep_firstlaunch_pmc.txt).  ONE experiment that
# changes TLB reach rather than timing: the five cls head tensors in one engine-owned, 2 MiB-aligned, persistently mapped buffer
# (odtk/fused.py: _cls_arena, ODTK_CLS_ARENA=1; the library convolutions write into it).  Both arms run with ODTK_CONV_ROUTE=library so that
# the producer of every level is the same kernel family and the arena costs no copy.
#   (1) bench.py, alternating arms: prefilter time in the step, images/s
#   (2) rocprofv3 --pmc (own passes, --kernel-trace only): UTCL1 translation misses and L2-TLB busy cycles of prefilter_scan_kernel
O=gpurun_out/prefilter_tlb; mkdir -p $O
export TMPDIR=/tmp ODTK_CONV_ROUTE=library
BENCH="python3 bench.py --gpus 1 --steps 40 --warmup 8 --no-other-configs --no-eager-leg --cpu-seconds 0"
for rep in 1 2 3; do
  for A in 0 1; do
    if [ $A = 1 ]; then export ODTK_CLS_ARENA=1; else unset ODTK_CLS_ARENA; fi
    ( timeout 400 $BENCH --detail-out $O/bench_arena${A}_$rep.detail.json ) > $O/bench_arena${A}_$rep.json 2> $O/bench_arena${A}_$rep.err
    python - <<P
import json
d = json.loads(open('$O/bench_arena${A}_$rep.json').read().strip().splitlines()[-1])
print('arena=$A rep $rep: %.1f img/s  %.3f ms  prefilter %.2f us (frac %.4f)  select %.2f  nms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'], d['kernels_avg_us']['select_decode_kernel'], d['kernels_avg_us']['nms_kernel']))
P
  done
done
run() {   # tag -- command
  local tag=$1; shift; shift
  timeout 300 rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum --kernel-trace -d $O/$tag -o pmc -- "$@" > $O/$tag.log 2>&1
  python tools/pmc_read.py $(find $O/$tag -name '*_results.db' | head -1) --match odtk --skip 4 2>&1 | grep "prefilter_scan\|select_decode\|nms_kernel" | awk '{print $1, $2, $6, $NF}' | sed 's/_ZN4odtk[0-9]*//; s/INS_.*kd//; s/ILi.*kd//' > $O/$tag.txt
  cat $O/$tag.txt
}
STEP="python bench.py --steps 12 --warmup 6 --cpu-seconds 0 --no-eager-leg --no-other-configs"
unset ODTK_CLS_ARENA;    echo "== no arena: tlb counters"; run step_tlb_arena0 -- $STEP
export ODTK_CLS_ARENA=1; echo "== arena: tlb counters";    run step_tlb_arena1 -- $STEP
find $O -name "*.db" -size +8M -delete

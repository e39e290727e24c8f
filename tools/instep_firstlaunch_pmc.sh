#!/bin/bash
# Why does the FIRST launch of each post-processing kernel behind the convolutions cost ~4 us more than the same launch repeated
# (tools/prefilter_instep_probe.py: step 48.1 / 30.0 / 25.4 us, second launch right behind it 43.8 / ~27 / ~21.4 = the figures of the
# kernels alone; 60 us of idle GPU in between does not help)?  Not the clock then.  Candidates a first launch pays and a repeat does
# not: address translation (the head tensors' pages are new to the L2 TLB), instruction fetch (the kernel's code has left the L2s).
# Counters of the three kernels in both contexts, separate rocprofv3 --pmc passes (--kernel-trace only):
#   tlb    : GRBM_UTCL2_BUSY (cycles the L2 TLB is busy, chip-wide, in the dispatch window) vs GRBM_GUI_ACTIVE, UTCL1 misses / multi-miss stalls
#   icache : SQC_ICACHE_REQ / HITS / MISSES / BUSY_CYCLES
O=gpurun_out/firstlaunch_pmc; mkdir -p $O
export TMPDIR=/tmp
run() {   # tag counters... -- command
  local tag=$1; shift
  local counters=()
  while [ "$1" != "--" ]; do counters+=("$1"); shift; done
  shift
  timeout 240 rocprofv3 --pmc "${counters[@]}" --kernel-trace -d $O/$tag -o pmc -- "$@" > $O/$tag.log 2>&1
  python tools/pmc_read.py $(find $O/$tag -name '*_results.db' | head -1) --match odtk --skip 4 2>&1 | grep "prefilter_scan\|select_decode\|nms_kernel" | awk '{print $1, $2, $6, $NF}' | sed 's/_ZN4odtk[0-9]*//; s/INS_.*kd//; s/ILi.*kd//' > $O/$tag.txt
  cat $O/$tag.txt
}
STEP="python bench.py --steps 12 --warmup 6 --cpu-seconds 0 --no-eager-leg --no-other-configs"
B2B="python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 12"
echo "== in step: tlb";      run step_tlb GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum -- $STEP
echo "== back to back: tlb"; run b2b_tlb  GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum -- $B2B
echo "== in step: icache";      run step_ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES -- $STEP
echo "== back to back: icache"; run b2b_ic  SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES -- $B2B
find $O -name "*.db" -size +8M -delete

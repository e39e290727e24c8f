#!/bin/bash
# the prefilter takes ~46 us inside the step and ~41 us back to back on the same tensors.  A counter-backed reason:
# L2 / memory-side counters of prefilter_scan_kernel in both contexts (separate rocprofv3 --pmc passes, --kernel-trace only).
#   in step      : python bench.py (12 steps)                 -- the launch follows the head towers' last convolutions
#   back to back : python tools/postproc_bench.py (12 calls)  -- the launch follows the previous call's nms / detect kernel
# Reading (MI355X_MICROARCH.md "boundary": + B / 6 TB/s when the predecessor leaves B bytes dirty): the convolutions leave up to
# 32 MiB of the 245 MB cls tensor dirty in the eight L2s; those lines go to memory while the prefilter streams -- its dispatch
# window then shows memory-side WRITE requests far above the 2.2 MB of keys it stores itself.
O=gpurun_out/prefilter_pmc; mkdir -p $O
export TMPDIR=/tmp
run() {   # tag counters... -- command
  local tag=$1; shift
  local counters=()
  while [ "$1" != "--" ]; do counters+=("$1"); shift; done
  shift
  timeout 240 rocprofv3 --pmc "${counters[@]}" --kernel-trace -d $O/$tag -o pmc -- "$@" > $O/$tag.log 2>&1
  python tools/pmc_read.py $(find $O/$tag -name '*_results.db' | head -1) --match prefilter_scan --skip 4 > $O/$tag.txt 2>&1
  cat $O/$tag.txt
}
STEP="python bench.py --steps 12 --warmup 6 --cpu-seconds 0 --no-eager-leg --no-other-configs"
B2B="python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch 8 --iters 12"
run step_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- $STEP
run b2b_tcc  TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -- $B2B
run step_wr WRITE_SIZE -- $STEP
run b2b_wr  WRITE_SIZE -- $B2B
find $O -name "*.db" -size +8M -delete

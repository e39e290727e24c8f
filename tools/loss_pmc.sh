#!/bin/bash
# Round 5, first GPU call on the loss forward: what saturates at 4.1-4.5 TB/s?  (30 us with no arithmetic,
# slower with more bytes in flight.)  Counters of the loss kernels in three separate rocprofv3 passes (never --pmc together with a
# hip / hsa / sys trace), ~15 s each; summaries under gpurun_out/loss_pmc/, copy what is to be judged into profiles/.
#   pass 1 (SQ):  where the waves' cycles go -- parked on s_waitcnt / barriers, issue-stalled, issuing
#   pass 2 (TCC): L2 hits / misses and the memory-side request count of the stream
#   pass 3 (TCC): FETCH_SIZE (costs 3 of the 4 TCC slots)
O=gpurun_out/loss_pmc; mkdir -p $O
export TMPDIR=/tmp
run() {   # tag counters...
  local tag=$1; shift
  timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d $O/$tag -o pmc -- python tools/loss_form_probe.py --pmc > $O/$tag.log 2>&1
  python tools/pmc_read.py $(find $O/$tag -name '*_results.db' | head -1) --match loss --skip 3 > $O/$tag.txt 2>&1
  tail -30 $O/$tag.txt
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
find $O -name "*.db" -size +8M -delete

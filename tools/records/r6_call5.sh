#!/bin/bash
# Round 6, GPU call 5: where the NMS's time goes on a heavy image of the trained detector (phase trace, both round forms), the
# batch loop's own phases on the bench heads; the prefilter TLB experiment (tools/prefilter_tlb.sh).
O=gpurun_out/r6c5; mkdir -p $O
export TMPDIR=/tmp
for C in 0 1; do
  ( ODTK_NMS_CHUNKS=$C timeout 300 python tools/nms_clustered_probe.py --first 10 ) > $O/nms_clustered_first10_chunks$C.txt 2>&1; echo "== ODTK_NMS_CHUNKS=$C"; grep -v amdgpu.ids $O/nms_clustered_first10_chunks$C.txt | grep -v "^ *[0-9]* |" | head -40
done
( ODTK_NMS_CHUNKS=0 timeout 300 python tools/nms_trace_probe.py ) > $O/nms_rn50_batched.txt 2>&1; grep -v amdgpu.ids $O/nms_rn50_batched.txt | grep "phases\|back to back, event\|img  0"
( time bash tools/prefilter_tlb.sh ) > $O/prefilter_tlb.txt 2>&1; cat $O/prefilter_tlb.txt

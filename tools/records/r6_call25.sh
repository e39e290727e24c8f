#!/bin/bash
# Round 6, GPU call 25: tools/loss_layout_probe.py with the walk and the reduce launch of the workspace form timed apart (ODTK_KERNEL_LOSS_REDUCE),
# the reduce launch with its loads in flight together, the atomics form with contiguous trips, the new backward defaults.
O=gpurun_out/r6c25; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python tools/loss_layout_probe.py ) > $O/loss_layout_probe.txt 2>&1
grep -v "amdgpu\|^agree.*ok$" $O/loss_layout_probe.txt | tail -70 | cut -c1-200

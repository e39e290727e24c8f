#!/bin/bash
# Round 6, GPU call 33: the loss kernels' counters again (tools/loss_pmc.sh = round 5's record profiles/r05_loss_pmc.txt) on the final
# kernels: does the backward's L2 request count show the box-delta walk's change (4.56 M requests before, 1.5 M of them its
# element-per-store writes)?
export TMPDIR=/tmp
( time bash tools/loss_pmc.sh ) > gpurun_out/loss_pmc_r06.txt 2>&1
grep -v amdgpu gpurun_out/loss_pmc_r06.txt | tail -60 | cut -c1-220

#!/bin/bash
# Round 6, GPU call 24: tools/loss_layout_probe.py once more (call 23 stopped at the workspace form: the tool's timing helper
# asserted one launch per call, the workspace form is two).
O=gpurun_out/r6c24; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python tools/loss_layout_probe.py ) > $O/loss_layout_probe.txt 2>&1
grep -v "amdgpu\|^agree.*ok$" $O/loss_layout_probe.txt | tail -70 | cut -c1-200

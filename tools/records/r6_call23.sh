#!/bin/bash
# Round 6, GPU call 23: the loss kernels' untested levers (per-wave partial sums, contiguous trips) and the backward's box-delta walk in
# memory order (tools/loss_layout_probe.py); tests/test_gpu_loss.py on the new default walk.
O=gpurun_out/r6c23; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python -m pytest tests/test_gpu_loss.py tests/test_targets.py -m gpu -q -x ) > $O/pytest_loss.txt 2>&1; grep -v amdgpu $O/pytest_loss.txt | tail -4
( time timeout 500 python tools/loss_layout_probe.py ) > $O/loss_layout_probe.txt 2>&1
grep -v "amdgpu\|^agree.*ok$" $O/loss_layout_probe.txt | tail -60 | cut -c1-200

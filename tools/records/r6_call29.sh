#!/bin/bash
# Round 6, GPU call 29: tests/test_conv_plan.py::test_plan_file_replays_across_processes failed once in call 28 (same plan hash,
# another head-tensor digest) after passing in every earlier call -> tools/plan_replay_probe.py: eight processes on one plan file,
# per-tensor and per-layer digests; then the test file itself three times.
O=gpurun_out/r6c29; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python tools/plan_replay_probe.py 8 ) > $O/plan_replay_probe.txt 2>&1; grep -v amdgpu $O/plan_replay_probe.txt | cut -c1-400 | tail -20
for i in 1 2 3; do ( timeout 300 python -m pytest tests/test_conv_plan.py -m gpu -q -x ) > $O/pytest_$i.txt 2>&1; grep -v amdgpu $O/pytest_$i.txt | tail -1; done

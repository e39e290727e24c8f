#!/bin/bash
# Round 6, GPU call 30: the plan-replay children under cudnn.deterministic (call 29: MIOpen's find-mode pick for the skip-input
# convolutions did not reproduce its own bits on that box) -- the probe both ways, then the WHOLE GPU suite without -x (what else
# depends on the box?) and smoke().
O=gpurun_out/r6c30; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python tools/plan_replay_probe.py 4 ) > $O/plan_replay_probe_find.txt 2>&1; grep -v amdgpu $O/plan_replay_probe_find.txt | cut -c1-200 | tail -3
( PROBE_DETERMINISTIC=1 timeout 300 python tools/plan_replay_probe.py 4 ) > $O/plan_replay_probe_deterministic.txt 2>&1; grep -v amdgpu $O/plan_replay_probe_deterministic.txt | cut -c1-200 | tail -3
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; grep -v amdgpu $O/pytest_gpu.txt | tail -8
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" ) > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt

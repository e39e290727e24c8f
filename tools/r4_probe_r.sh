#!/bin/bash
# round 4, probe r: the two tests the stale compiled binding failed in the final run, with the rebuilt module
export TMPDIR=/tmp
mkdir -p gpurun_out/r4r
timeout 300 python -m pytest tests/test_compiled_binding.py tests/test_gpu_threads.py -q -m gpu > gpurun_out/r4r/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4r/pytest.log

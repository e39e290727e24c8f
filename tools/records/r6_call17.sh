#!/bin/bash
# Round 6, GPU call 17: long differential fuzz of the axis-aligned NMS on detector-like clusters in both input forms
# (tools/nms_fuzz_long.py, 1500 cases) + the fuzz suite with its new cases.
O=gpurun_out/r6c17; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python tools/nms_fuzz_long.py --seeds 0:1500 ) > $O/nms_fuzz_long.txt 2>&1; grep -v amdgpu.ids $O/nms_fuzz_long.txt | tail -12 | cut -c1-300
( time timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q ) > $O/pytest_fuzz.txt 2>&1; tail -4 $O/pytest_fuzz.txt

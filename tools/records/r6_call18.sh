#!/bin/bash
# Round 6, GPU call 18: tools/nms_fuzz_long.py on 38 500 more cases (seeds 1500 .. 39999).
O=gpurun_out/r6c18; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python tools/nms_fuzz_long.py --seeds 1500:40000 ) > $O/nms_fuzz_long.txt 2>&1; grep -v amdgpu.ids $O/nms_fuzz_long.txt | tail -12 | cut -c1-300

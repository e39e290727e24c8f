#!/bin/bash
# Round 6, GPU call 19: tools/decode_fuzz_long.py, 2500 cases (levels up to 128 x 160 cells: shared segments, plateaus, ties).
O=gpurun_out/r6c19; mkdir -p $O
export TMPDIR=/tmp
( time timeout 2400 python tools/decode_fuzz_long.py --seeds 0:2500 ) > $O/decode_fuzz_long.txt 2>&1; grep -v amdgpu.ids $O/decode_fuzz_long.txt | tail -12 | cut -c1-400

#!/bin/bash
# Round 6, GPU call 22: long differential fuzz of the FUSED fast path (raw 16-bit logits, channels_last, folded biases, threshold
# table) against the strict op fed with what the reference pipeline materialises first (tools/fused_fuzz_long.py).
O=gpurun_out/r6c22; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python tools/fused_fuzz_long.py --seeds 0:4000 ) > $O/fused_fuzz_long.txt 2>&1
grep -v amdgpu $O/fused_fuzz_long.txt | tail -8 | cut -c1-600

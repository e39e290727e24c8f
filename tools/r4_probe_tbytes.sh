#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
for b in 2 4 8 16 32; do
  timeout 200 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --bias --batch $b --iters 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('batch', d['batch'], 'bytes', d['alg_bytes'], 'prefilter_us', d['kernels_us']['prefilter_scan_kernel'], 'select_us', d['kernels_us']['select_decode_kernel'], 'nms_us', d['kernels_us']['nms_kernel'], 'GBps', d['prefilter_GBps'])" | tee -a $O/tbytes.txt
done
for b in 8 32; do
  timeout 200 python tools/postproc_bench.py --kind sparse --dtype bf16 --logits --channels-last --batch $b --iters 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('NO BIAS batch', d['batch'], 'bytes', d['alg_bytes'], 'prefilter_us', d['kernels_us']['prefilter_scan_kernel'], 'GBps', d['prefilter_GBps'])" | tee -a $O/tbytes.txt
done

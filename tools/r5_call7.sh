#!/bin/bash
# Round 5, GPU call 7: what the step is made of now (kernel trace of the bench command), the AP proxy on three seeds
O=gpurun_out/r5c7; mkdir -p $O
export TMPDIR=/tmp
STEPS=40
rocprofv3 --kernel-trace --stats -d $O/bench -o bench -- python bench.py --steps $STEPS --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
DB=$(find $O/bench -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" --steady prefilter_scan:$STEPS --csv $O/bench_steady_kernel_stats.csv --top 45 > $O/bench_steady_kernel_stats.txt 2>&1
head -50 $O/bench_steady_kernel_stats.txt
find $O -name "*.db" -size +8M -delete
( time timeout 600 python tools/detection_ap_seeds.py --seeds 0 1 2 ) > $O/detection_ap_seeds.txt 2>&1; tail -14 $O/detection_ap_seeds.txt

#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_targets.py tests/test_gpu_fused_model.py -x -q -m gpu > $O/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a $O/summary.txt
tail -12 $O/tests1.log
timeout 500 python bench.py --mode train --rotated-bbox --steps 8 --warmup 3 > $O/train_rot.json 2> $O/train_rot.err; echo "train rot rc=$?" | tee -a $O/summary.txt
tail -3 $O/train_rot.err; cat $O/train_rot.json | head -c 1500

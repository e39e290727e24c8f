#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_threads.py tests/test_gpu_loss.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests1.log 2>&1; echo "tests1 rc=$?" | tee -a $O/summary.txt
tail -15 $O/tests1.log
timeout 400 python tools/prefilter_instep_probe.py > $O/instep.txt 2>&1
tail -12 $O/instep.txt

#!/bin/bash
# Round 5, GPU call 10: the round's profiles (rocprofv3 kernel stats of the bench command, PMC traffic, MFMA utilisation), then the
# driver's own bench command
O=gpurun_out/r5c10; mkdir -p $O
export TMPDIR=/tmp
( time bash tools/profile_round.sh r05 ) > $O/profile_round.txt 2>&1; tail -5 $O/profile_round.txt
head -45 gpurun_out/prof_r05/r05_bench_steady_kernel_stats.txt
cat gpurun_out/prof_r05/r05_pmc_traffic.json | head -30
tail -12 gpurun_out/prof_r05/r05_pmc_mfma_bench.txt
( time timeout 120 python -m pytest "tests/test_gpu_loss.py::test_model_training_step_fused_loss_equals_torch_loss" -q ) 2>&1 | tail -4

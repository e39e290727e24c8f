#!/bin/bash
# round 4: the full GPU test suite, the default bench line with its legs, and the round's profiles
O=gpurun_out/r4final; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt; tail -2 $O/smoke.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
tail -c 600 $O/bench_default.json | head -c 300; echo
STEPS=50 timeout 1500 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; echo "profiles rc=$?" | tee -a $O/summary.txt
tail -5 $O/profile_round.log
cat $O/summary.txt

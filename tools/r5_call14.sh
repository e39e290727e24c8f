#!/bin/bash
# Round 5, GPU call 14: the whole GPU suite on the final tree (cooperative selection route on), the selection suites again with
# every barrier timing out (ODTK_SELECT_COOP_TICKS=1: mixed routes), smoke
O=gpurun_out/r5c14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu_all.txt 2>&1; tail -5 $O/pytest_gpu_all.txt
( time ODTK_SELECT_COOP_TICKS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_threads.py tests/test_gpu_rotated.py tests/test_gpu_nms_corners.py -q ) > $O/pytest_ticks1.txt 2>&1; tail -4 $O/pytest_ticks1.txt
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; grep smoke $O/smoke.txt

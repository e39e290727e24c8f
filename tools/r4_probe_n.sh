#!/bin/bash
O=gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/rotated_train_probe.py > $O/rot_train.txt 2>&1; tail -16 $O/rot_train.txt
timeout 400 python tools/graph_ab_bs8.py > $O/graph_ab.txt 2>&1; tail -4 $O/graph_ab.txt

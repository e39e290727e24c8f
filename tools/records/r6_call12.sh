#!/bin/bash
# Round 6, GPU call 12: the trained-detector AP on SIX training seeds and 1024 held-out scenes each: is the bf16 engine's
# deviation from the reference (+0.0032 / -0.0009 on two seeds x 256 scenes, call 1) a bias or scatter?
O=gpurun_out/r6c12; mkdir -p $O
export TMPDIR=/tmp
( time timeout 3300 python tools/trained_ap.py --seeds 0 1 2 3 4 5 --iterations 2500 --images 1024 --json $O/trained_ap_6seeds.json ) > $O/trained_ap_6seeds.txt 2> $O/trained_ap_6seeds.err; head -12 $O/trained_ap_6seeds.txt | cut -c1-400; tail -3 $O/trained_ap_6seeds.err

#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root:  bash tools/profile_round.sh r02
# Everything lands under gpurun_out/prof_<round>/ ; copy what is to be judged into profiles/.
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command -> steady-state per-kernel CSV
#      (its average prefilter duration must agree with the hipEvent figure in the bench line)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) on the
#      post-processing alone, bf16 fused (what the step runs) and fp32 (the reference boundary) -> traffic JSON
#   3. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on a short bench run -> MFMA utilisation of the convolutions
R=${1:-r05}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
STEPS=${STEPS:-50}

rocprofv3 --kernel-trace --stats -d $OUT/bench -o bench -- python bench.py --steps $STEPS --warmup 10 --cpu-seconds 0 --no-eager-leg --no-other-configs \
    > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
DB=$(find $OUT/bench -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" --steady prefilter_scan:$STEPS --csv $OUT/${R}_bench_steady_kernel_stats.csv --top 25 > $OUT/${R}_bench_steady_kernel_stats.txt 2>&1

echo '{"_comment": "HBM traffic of prefilter_scan_kernel per launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md, + WRITE_SIZE; KB = 1024 B). Sources: profiles/'$R'_pmc_postproc_*.csv. bench.py quotes these for the matching workload (bs=8, 800x1280, A=9, C=80)."}' > $OUT/${R}_pmc_traffic.json
pmc() {   # name key bytes-per-score args...
  local name=$1 key=$2 bps=$3; shift 3
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_${name}_$c -o pmc -- python tools/postproc_bench.py "$@" --batch 8 --iters 5 > $OUT/pmc_${name}_$c.log 2>&1
  done
  python tools/pmc_traffic.py "$(find $OUT/pmc_${name}_FETCH_SIZE -name '*_results.db' | head -1)" "$(find $OUT/pmc_${name}_WRITE_SIZE -name '*_results.db' | head -1)" \
      --csv $OUT/${R}_pmc_postproc_${name}_bs8.csv --json $OUT/${R}_pmc_traffic.json --key $key --scores 122860800 --bytes-per-score $bps \
      --comment "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) -- python tools/postproc_bench.py $* --batch 8 --iters 5; KB as reported, FETCH_SIZE x2 on gfx950" > $OUT/pmc_${name}.txt 2>&1
}
pmc bf16_fused_sparse bf16_logits_channels_last 2 --kind sparse --dtype bf16 --logits --channels-last --bias
pmc fp32_sparse fp32_scores_nchw 4 --kind sparse

# 3. MFMA utilisation of the convolution kernels (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, one pass, short run)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 6 --warmup 4 \
    --cpu-seconds 0 --no-eager-leg --no-other-configs > $OUT/pmc_mfma.log 2>&1
python tools/pmc_mfma.py "$(find $OUT/pmc_mfma -name '*_results.db' | head -1)" --steps 3 --csv $OUT/${R}_pmc_mfma_bench.csv > $OUT/${R}_pmc_mfma_bench.txt 2>&1
# (the other BASELINE configurations are legs of the default bench.py run since round 3: `other_configs` on its line)
rm -rf $OUT/bench/*/*.db.tmp 2>/dev/null
# the raw rocpd databases are large: keep only the summaries
find $OUT -name "*.db" -size +8M -delete
ls -la $OUT

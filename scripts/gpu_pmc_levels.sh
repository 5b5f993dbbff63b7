#!/bin/bash
# What the lone wave per SIMD waits for: average outstanding VMEM / SMEM / LDS instructions per wave (SQ_INST_LEVEL_* over
# SQ_WAVE_CYCLES) in one single-bound solve of the one-worker-per-SIMD configuration.
# usage: scripts/gpu_pmc_levels.sh <tag> SIZE K SECONDS
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | tr ' ' '_' | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_$N -- python3 $ROOT/scripts/gpu_rung.py $2 $3 $4 "workers=1024,ramp=-1,slice_ms=250" > $OUT/${TAG}_$N.log 2>&1 || { tail -3 $OUT/${TAG}_$N.log; continue; }
  grep -E "Unsat|Interrupted|Sat" $OUT/${TAG}_$N.log | tail -1 | cut -c1-120
  python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/prof_${TAG}_$N -name "*counter_collection.csv") | tee $OUT/${TAG}_pmc_$N.json
done

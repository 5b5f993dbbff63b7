#!/bin/bash
# PMC passes over one single-bound solve in the one-worker-per-SIMD configuration (1024 workers, LDS builds).
# usage: scripts/gpu_pmc_rung.sh <tag> SIZE K SECONDS
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_$N -- python3 $ROOT/scripts/gpu_rung.py $2 $3 $4 "workers=1024,ramp=-1,slice_ms=250" > $OUT/${TAG}_$N.log 2>&1 || { tail -3 $OUT/${TAG}_$N.log; exit 1; }
  grep -E "Unsat|Interrupted|Sat" $OUT/${TAG}_$N.log | tail -1
  python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/prof_${TAG}_$N -name "*counter_collection.csv") | tee $OUT/${TAG}_pmc_$N.json
done

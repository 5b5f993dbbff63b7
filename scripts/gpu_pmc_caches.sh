#!/bin/bash
# Cache side of one single-bound solve in the one-worker-per-SIMD configuration: L2 hits / misses and the vector L1's
# accesses against what it passes on to L2.
# usage: scripts/gpu_pmc_caches.sh <tag> SIZE K SECONDS
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_$N -- python3 $ROOT/scripts/gpu_rung.py $2 $3 $4 "workers=1024,ramp=-1,slice_ms=250" > $OUT/${TAG}_$N.log 2>&1 || { tail -3 $OUT/${TAG}_$N.log; continue; }
  grep -E "Unsat|Interrupted|Sat" $OUT/${TAG}_$N.log | tail -1 | cut -c1-120
  python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/prof_${TAG}_$N -name "*counter_collection.csv") | tee $OUT/${TAG}_pmc_$N.json
done

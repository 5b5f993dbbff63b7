#!/bin/bash
# Instruction-fetch side of one single-bound solve in the one-worker-per-SIMD configuration (the search kernel is 94 KB of
# code): I-cache requests / misses and the share of cycles waves wait for an instruction.
# usage: scripts/gpu_pmc_icache.sh <tag> SIZE K SECONDS
TAG=$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/${TAG}_avail.txt 2>&1
grep -o -E "SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQ_WAIT_INST[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQC_INST[A-Z_]*" $OUT/${TAG}_avail.txt | sort -u | tr '\n' ' '; echo
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH"; do
  N=$(echo $C | tr ' ' '_' | tr 'A-Z' 'a-z')
  rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_$N -- python3 $ROOT/scripts/gpu_rung.py $2 $3 $4 "workers=1024,ramp=-1,slice_ms=250" > $OUT/${TAG}_$N.log 2>&1 || { tail -3 $OUT/${TAG}_$N.log; continue; }
  grep -E "Unsat|Interrupted|Sat" $OUT/${TAG}_$N.log | tail -1 | cut -c1-120
  python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/prof_${TAG}_$N -name "*counter_collection.csv") | tee $OUT/${TAG}_pmc_$N.json
done

#!/bin/bash
# A/B of two builds of the solver library on the GPU box: fixed-time rungs (conflicts/s of 1024 workers) and the 64x64 sweep (prop/s).
# usage: scripts/gpu_ab.sh TAG "libA.so libB.so ..." [RUNG_SECONDS]   ("" = the in-tree product library)
TAG=$1; LIBS=$2; SEC=${3:-8}; OUT=gpurun_out/${TAG}_ab.log; : > $OUT
for L in $LIBS; do
  [ "$L" = "product" ] && unset BENCH_LIB || export BENCH_LIB=$L
  echo "== $L" >> $OUT
  timeout -k 10 100 python3 scripts/gpu_rung.py 28 11 $SEC "workers=1024" 2>&1 | grep -E "Interrupted|Unsat|Sat" | cut -c1-200 >> $OUT || exit 1
  timeout -k 10 100 python3 scripts/gpu_rung.py 32 14 $SEC "workers=1024" 2>&1 | grep -E "Interrupted|Unsat|Sat" | cut -c1-200 >> $OUT || exit 1
  timeout -k 10 200 python3 bench.py --no-cpu --steps 8 --warmup 2 --first-unsat-sizes "" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sweep prop/s %.4g frac %.4f' % (d['value'], d['roofline']['frac']))" >> $OUT || exit 1
done
cat $OUT

#!/bin/bash
# Runs on the GPU box (through gpurun): the bench line, the kernel trace of the same command and the PMC passes the
# roofline's `traffic` comes from; summaries land in gpurun_out/<tag>_* (copy the ones to keep into profiles/).
# usage: scripts/gpu_profile_bench.sh <tag> [extra bench.py args...]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profile] bench line"; python3 $ROOT/bench.py "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
cat $OUT/${TAG}_bench.json | head -c 600; echo
echo "[profile] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/${TAG}_kt.log 2>&1 || { tail -5 $OUT/${TAG}_kt.log; exit 1; }
cp $(find $OUT/prof_${TAG}_kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
head -3 $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | tr 'A-Z' 'a-z')
  echo "[profile] pmc $C"
  rocprofv3 --pmc $C --output-format csv -d $OUT/prof_${TAG}_$N -- python3 $ROOT/bench.py --no-cpu "$@" > $OUT/${TAG}_pmc_$N.log 2>&1 || { tail -5 $OUT/${TAG}_pmc_$N.log; exit 1; }
  python3 $ROOT/scripts/summarize_pmc.py $(find $OUT/prof_${TAG}_$N -name "*counter_collection.csv") > $OUT/${TAG}_pmc_$N.json
  cat $OUT/${TAG}_pmc_$N.json
done
# the traffic summary bench.py attaches to its roofline - bound to the build it was measured on
python3 - <<PY
import csv, glob, json, subprocess
ident = json.loads(subprocess.run(["python3", "$ROOT/bench.py", "--build-identity"], capture_output=True, text=True).stdout)
f = json.load(open("$OUT/${TAG}_pmc_fetch_size.json"))["FETCH_SIZE"]
w = json.load(open("$OUT/${TAG}_pmc_write_size.json"))["WRITE_SIZE"]
names = [r["Name"] for r in csv.DictReader(open("$OUT/${TAG}_kernel_stats.csv")) if "ms_search_kernel" in r["Name"]]
out = {"kernel": names[0] if names else None, "git_head": ident["git_head"], "kernel_source_sha16": ident["kernel_source_sha16"],
       "command": "python3 bench.py --no-cpu $*", "launches": [f["launches"], w["launches"]],
       "fetch_size_kb_per_launch": f["per_launch"], "write_size_kb_per_launch": w["per_launch"],
       "hbm_bytes_per_launch": (f["per_launch"] + w["per_launch"]) * 1024,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of the command above (all its launches of the kernel "
                 "averaged, warm-up included); KB -> bytes x1024; counts L2<->fabric traffic, i.e. HBM plus Infinity-Cache hits; "
                 "FETCH_SIZE calibration for this access pattern: ${TAG}_chase_calibration.txt"}
json.dump(out, open("$OUT/${TAG}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
echo "[profile] FETCH_SIZE calibration on random 64-byte lines (scripts/microbench/chase.hip)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/chase $ROOT/scripts/microbench/chase.hip && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_${TAG}_chase -- /tmp/chase 8192 1024 2000 > $OUT/${TAG}_chase.log 2>&1
cat $OUT/${TAG}_chase.log | tail -3
python3 - <<PY
import csv, glob
rows = [r for f in glob.glob("$OUT/prof_${TAG}_chase/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "chase" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
# the timed launch is the one with the most traffic (the warm-up makes 100 steps)
v = max(float(r["Counter_Value"]) for r in rows) * 1024
exp = 1024 * 64 * 2000 * 64
print(f"chase: FETCH_SIZE {v:.4g} B for {exp:.4g} B of random 64-byte lines: factor {exp / v:.3f}")
open("$OUT/${TAG}_chase_calibration.txt", "w").write(f"FETCH_SIZE {v:.6g} bytes reported for {exp} bytes of dependent random 64-byte-line loads (8 GiB buffer, 1024 waves x 64 chains x 2000 steps): true/reported = {exp / v:.4f}\n")
PY

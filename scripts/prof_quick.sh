#!/bin/bash
# quick rocprofv3 passes of an arbitrary python script: kernel trace + FETCH/WRITE/TCC counters.
# usage (on the GPU box): scripts/prof_quick.sh <tag> <script.py> [args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/q_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
PASSES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU")
[ -n "${ONLY_TRACE:-}" ] && PASSES=()
for pmc in "${PASSES[@]}"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-30)
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o p -- python3 $ROOT/$@ > $OUT/pmc_$name.log 2>&1
done
find $OUT -name "*.db" -delete
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in list(csv.DictReader(open(f))):
        if row["Name"].startswith("rn::") or row["Name"].startswith("void rn::"):
            print(row["Name"][:60], row["Calls"], "avg_us", float(row["AverageNs"])/1e3)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if "wgrad" in k[0] or "level" in k[0]:
            print(k, len(v), sum(v)/len(v))
PY
find $OUT -size +4M -delete

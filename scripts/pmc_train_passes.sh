#!/bin/bash
# rocprofv3 --pmc passes over a few training steps of one chain mode (scripts/pmc_train.py), one counter group per pass:
#   bash scripts/pmc_train_passes.sh <tag> <mode> "<counters pass 1>" "<counters pass 2>" ...
# prints, per kernel of this library, the LAST launch's value of every counter (the launches of a step repeat)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; shift 2
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmct_$TAG/p$i --output-format csv -- python3 scripts/pmc_train.py $MODE > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmct_$TAG/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "rn::" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(k)
    for c, v in sorted(d[k].items()):
        print(f"    {c:34s} last {v[-1]:.6g}   mean {sum(v) / len(v):.6g}   (launches {len(v)})")
PY
find gpurun_out/pmct_$TAG -name "*.db" -delete
find gpurun_out/pmct_$TAG -size +4M -delete

#!/bin/bash
# rocprofv3 --pmc passes over scripts/pmc_one.py (one counter group per pass, as the guide prescribes):
#   bash scripts/pmc_passes.sh <tag> <lib.so | -> <precision> "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; LIB=$2; PREC=$3; shift 3
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/pmc_$TAG/p$i --output-format csv -- python3 scripts/pmc_one.py $LIB $PREC > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_$TAG/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "level_fwd" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(f"{k:36s} {v[-1]:.5g}   (launches {len(v)})")
PY

#!/bin/bash
# FETCH_SIZE / TCC hit counters of the weight-gradient GEMM for several library builds: bash scripts/pmc_wgrad_ab.sh ab/a.so ab/b.so ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the variants are copied over the in-tree library (rocprofv3 must start python itself: no env / wrapper hop): the original comes
# back when the script ends, however it ends
cp refnerf-pl_amd/csrc/librefnerf_hip.so /tmp/librefnerf_hip.intree.so
trap 'cp /tmp/librefnerf_hip.intree.so refnerf-pl_amd/csrc/librefnerf_hip.so' EXIT
for lib in "$@"; do
  cp $lib refnerf-pl_amd/csrc/librefnerf_hip.so
  tag=$(basename $lib .so)
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | tr ' ' '+')
    rocprofv3 --pmc $grp -d gpurun_out/pmc_wg_$tag/$name --output-format csv -- python3 scripts/pmc_train.py f16x2 > /dev/null 2>&1
  done
  python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmc_wg_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_f16s" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: "%.4g" % (sum(v) / len(v)) for k, v in d.items()})
PY
done

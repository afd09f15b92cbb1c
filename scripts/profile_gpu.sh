#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel trace + PMC passes.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries to profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-image --no-other-configs ${BENCH_ARGS:-}"
# counters do not depend on the clock state: short runs keep the per-dispatch CSVs small
PARGS="$ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-image --no-other-configs ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/trace.log 2>&1
for pmc in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pmc | tr ' ' '+' | cut -c1-40)
  rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $PARGS > $OUT/pmc_$name.log 2>&1
  # keep the rows of this library's kernels only (the ATen kernels of the losses / optimiser are 90 % of the file)
  python3 - "$OUT/pmc_$name/pmc_counter_collection.csv" <<'PY'
import csv, sys
p = sys.argv[1]
rows = list(csv.reader(open(p)))
k = rows[0].index("Kernel_Name")
keep = [rows[0]] + [r for r in rows[1:] if "rn::" in r[k]]
csv.writer(open(p, "w", newline="")).writerows(keep)
PY
done
find $OUT -name "*.db" -delete
find $OUT -size +8M -delete
ls -R $OUT | head -60
du -sh $OUT

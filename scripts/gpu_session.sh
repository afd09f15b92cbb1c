#!/bin/bash
# One GPU-box session: GPU tests (all, not -x), smoke(), bench.py in every configuration, the 2-rank launcher smoke.
# Outputs under gpurun_out/session_<tag>/.
TAG=${1:-s}
OUT=gpurun_out/session_$TAG
mkdir -p $OUT
python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -40 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_C2.json 2> $OUT/bench_C2.err; echo "bench C2 rc=$?"; cp gpurun_out/bench_full.json $OUT/bench_C2.full.json; grep -v "^BENCH_FULL" $OUT/bench_C2.err > $OUT/bench_C2.err.txt; rm -f $OUT/bench_C2.err
python bench.py --config C3 --steps 20 --warmup 5 > $OUT/bench_C3.json 2> $OUT/bench_C3.err; echo "bench C3 rc=$?"; cp gpurun_out/bench_full.json $OUT/bench_C3.full.json; grep -v "^BENCH_FULL" $OUT/bench_C3.err > $OUT/bench_C3.err.txt; rm -f $OUT/bench_C3.err
python bench.py --config C4 --steps 20 --warmup 5 > $OUT/bench_C4.json 2> $OUT/bench_C4.err; echo "bench C4 rc=$?"; cp gpurun_out/bench_full.json $OUT/bench_C4.full.json; grep -v "^BENCH_FULL" $OUT/bench_C4.err > $OUT/bench_C4.err.txt; rm -f $OUT/bench_C4.err
python bench.py --config C5 --steps 4 --warmup 1 > $OUT/bench_C5.json 2> $OUT/bench_C5.err; echo "bench C5 rc=$?"; cp gpurun_out/bench_full.json $OUT/bench_C5.full.json; grep -v "^BENCH_FULL" $OUT/bench_C5.err > $OUT/bench_C5.err.txt; rm -f $OUT/bench_C5.err
bash scripts/smoke_two_ranks.sh --config C4; cp gpurun_out/n2.log $OUT/n2_C4.log
bash scripts/smoke_two_ranks.sh --config C5 --rays 1024 --steps 3; cp gpurun_out/n2.log $OUT/n2_C5.log
for f in $OUT/bench_*.json; do echo "== $f"; cut -c1-600 $f; done
for f in $OUT/bench_*.err.txt; do tail -n 2 $f; done

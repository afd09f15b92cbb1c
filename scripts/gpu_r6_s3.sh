#!/bin/bash
# round 6, session 3: the whole GPU suite after the parity / ABI changes
OUT=gpurun_out/r6_s3
mkdir -p $OUT
rm -f gpurun_out/parity_full_size.json
python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" $OUT/pytest.log | tail -n 25
cp gpurun_out/parity_full_size.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 5 $OUT/smoke.log

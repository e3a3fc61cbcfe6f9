#!/bin/bash
# round 4, second session, GPU call 4: whole-block candidates for the aligned form; the half sweep's trailing update back and forth (update_rev)
OUT=$PWD/gpurun_out/${1:-r04b_4}; mkdir -p $OUT; REPO=$PWD
timeout 500 python -m pytest tests/test_sparse_contraction.py tests/test_device_group.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -5 $OUT/pytest_sparse.txt
timeout 300 python scripts/sweep_option_ab.py update_rev 0 1 0 1 --configs=5,2,4 > $OUT/update_rev_ab.txt 2>&1; cat $OUT/update_rev_ab.txt
for rv in 0 1; do MIK_UPDATE_REV=$rv timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --no-other --config 5 > $OUT/bench_c5_rev$rv.json 2> $OUT/bench_c5_rev$rv.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c5_rev$rv.json").read().strip().split("\n")[-1])
print("update_rev $rv", d["value"], d["ms_per_step"], d["phases_ms_per_step"])
EOF
done

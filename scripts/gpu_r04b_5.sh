#!/bin/bash
# round 4, second session, GPU call 5: moving window over shuffled points (device sort), cached station order, update_rev auto
OUT=$PWD/gpurun_out/${1:-r04b_5}; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_sparse_contraction.py -m gpu -x -q --tb=short -k "moving or window or sparse or sorted or mw" > $OUT/pytest_mw.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_mw.txt; tail -6 $OUT/pytest_mw.txt
timeout 200 python scripts/mw_sorted_ab.py > $OUT/mw_sorted_ab.txt 2>&1; cat $OUT/mw_sorted_ab.txt
timeout 200 python scripts/execute_breakdown.py 5 2 > $OUT/execute_breakdown.txt 2>&1; cat $OUT/execute_breakdown.txt
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --no-other --config 5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c5.json").read().strip().split("\n")[-1])
print("config 5", d["value"], d["ms_per_step"], d["phases_ms_per_step"], d["host_overhead"]["ms_per_step"])
EOF

#!/bin/bash
# round 4, second session, GPU call 3: per-K-tile candidates of k_rhs, group members inherit the sparse options; config 5 with one and two launch lanes
OUT=$PWD/gpurun_out/${1:-r04b_3}; mkdir -p $OUT; REPO=$PWD
timeout 500 python -m pytest tests/test_sparse_contraction.py tests/test_device_group.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -5 $OUT/pytest_sparse.txt
timeout 300 python scripts/sparse_rows_ab.py --quick > $OUT/sparse_rows_ab.txt 2>&1; echo "exit $?" >> $OUT/sparse_rows_ab.txt; cat $OUT/sparse_rows_ab.txt
for ln in 1 2; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --no-other --config 5 --sparse-lanes $ln > $OUT/bench_c5_lanes$ln.json 2> $OUT/bench_c5_lanes$ln.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c5_lanes$ln.json").read().strip().split("\n")[-1])
print("lanes $ln", d["value"], d["ms_per_step"], d["phases_ms_per_step"])
EOF
done
MIK_FUZZ_CASES=400 timeout 300 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -4 > $OUT/fuzz400.txt; cat $OUT/fuzz400.txt

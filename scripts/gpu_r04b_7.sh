#!/bin/bash
# round 4, second session, GPU call 7: auto mode of the point sort leaves small jobs alone
OUT=$PWD/gpurun_out/${1:-r04b_7}; mkdir -p $OUT
timeout 400 python -m pytest tests/test_sparse_contraction.py tests/test_device_group.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -3 $OUT/pytest_sparse.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
timeout 200 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1; cat $OUT/small_problem_latency.txt
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --pmc off --no-other --config 5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-330 $OUT/bench_c5.json | cut -c100-330

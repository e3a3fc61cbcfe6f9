#!/bin/bash
# round 4, second session, GPU call 10: the whole GPU suite, smoke() and the default bench run at HEAD
OUT=$PWD/gpurun_out/${1:-r04b_10}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
( time timeout 600 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-200 $OUT/bench_c2.json; tail -4 $OUT/bench_c2.err

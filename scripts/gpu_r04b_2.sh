#!/bin/bash
# round 4, second session, GPU call 2: the points of every launch in Hilbert order (k_ps_*) -- parity, A/B, bench config 5
OUT=$PWD/gpurun_out/${1:-r04b_2}; mkdir -p $OUT; REPO=$PWD
timeout 500 python -m pytest tests/test_sparse_contraction.py tests/test_device_group.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -15 $OUT/pytest_sparse.txt
timeout 500 python scripts/sparse_rows_ab.py > $OUT/sparse_rows_ab.txt 2>&1; echo "exit $?" >> $OUT/sparse_rows_ab.txt; cat $OUT/sparse_rows_ab.txt
for sp in 0 1; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --no-other --config 5 --sort-points $sp > $OUT/bench_c5_sort$sp.json 2> $OUT/bench_c5_sort$sp.err; cut -c1-330 $OUT/bench_c5_sort$sp.json; done

#!/bin/bash
# round 4, second session, GPU call 6: degenerate point lists through the sorter; smoke()
OUT=$PWD/gpurun_out/${1:-r04b_6}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_sparse_contraction.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -5 $OUT/pytest_sparse.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt

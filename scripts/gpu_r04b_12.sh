#!/bin/bash
# round 4, second session, GPU call 12: two launch lanes through the gathered + sorted contraction
OUT=$PWD/gpurun_out/${1:-r04b_12}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_sparse_contraction.py -m gpu -x -q --tb=short > $OUT/pytest.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest.txt; tail -12 $OUT/pytest.txt

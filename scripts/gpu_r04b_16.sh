#!/bin/bash
# round 4, second session, GPU call 16: the whole GPU suite at the final HEAD
OUT=$PWD/gpurun_out/${1:-r04b_16}; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2

#!/bin/bash
# round 4, second session, GPU call 15: moving window over shuffled points with 131 072-point sort segments
OUT=$PWD/gpurun_out/${1:-r04b_15}; mkdir -p $OUT
timeout 200 python scripts/mw_sorted_ab.py > $OUT/mw_sorted_ab.txt 2>&1; cat $OUT/mw_sorted_ab.txt
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q --tb=short -k "moving or window or mw" 2>&1 | tail -2

#!/bin/bash
# Round-5 GPU runs, one parameterised script (the review asked for one instead of sixteen one-shots).
#   bash scripts/gpu_r05.sh STAGE [TAG]      -> output under gpurun_out/r05_TAG/
# STAGES
#   k2      the sweep tests + A/B of the deep trailing update (update_deep, update_tpb) at bench configs 3, 4, 2, 5
#   largen  parity beyond N = 8000 against the oracle (tests/test_large_n.py with MIK_SLOW_TESTS=1)
#   tests   the whole GPU suite
#   bench   bench.py at config 2 (default run, incl. other_configs + CPU leg + live PMC)
#   evidence  tests + bench + rocprofv3 kernel stats of the bench (profiles for the round)
STAGE=${1:-k2}; TAG=${2:-$STAGE}; OUT=$PWD/gpurun_out/r05_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
case $STAGE in
k2)
  timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sweep or panel_stream or inverse" > $OUT/pytest_k2.txt 2>&1; tail -3 $OUT/pytest_k2.txt
  for opt in "update_deep 0 1" "update_tpb 1 2 3 4 6 8 16"; do
    timeout 900 python scripts/sweep_option_ab.py $opt 2>&1 | tee -a $OUT/update_deep_ab.txt
  done
  ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
  ;;
largen)
  MIK_SLOW_TESTS=1 timeout 1500 python -m pytest tests/test_large_n.py -m gpu -q -s > $OUT/pytest_large_n.txt 2>&1; grep -E "^n[0-9]|passed|failed|Error" $OUT/pytest_large_n.txt
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 1500 $OUT/bench_c2.json
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac

#!/bin/bash
# One gpurun call's worth of GPU work: environment probe, fp64 microbenchmark, GPU parity tests, bench, rocprof.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
REPO=$PWD
{ ls /root/reference 2>&1 | head -3; nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; } > $OUT/env.txt 2>&1
if [ -z "$SKIP_UBENCH" ]; then timeout 120 ./tools/ubench_f64 > $OUT/ubench_f64.txt 2>&1; fi
if [ -z "$SKIP_TESTS" ]; then timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --tb=short ${PYTEST_ARGS} > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; fi
if [ -z "$SKIP_BENCH" ]; then timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err; fi
if [ -z "$SKIP_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu ${BENCH_ARGS} > $OUT/prof_bench.json 2> $OUT/prof.err
  cd $REPO
  find $OUT/prof -name "*stats*" | head; 
fi
tail -5 $OUT/pytest_gpu.txt; cat $OUT/bench.json; tail -3 $OUT/bench.err

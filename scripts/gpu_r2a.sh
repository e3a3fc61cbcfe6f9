#!/bin/bash
# Round-2 first GPU pass: GPU tests (incl. device groups aliased onto the one GPU, full-size reference fixtures),
# the 1-GPU bench line (live PMC traffic + CPU protocol), launcher-free multi-device bench lines.
TAG=${1:-r2a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
{ nproc; free -g | head -2; rocm-smi --showproductname 2>&1 | head -8; grep -m1 "model name" /proc/cpuinfo; } > $OUT/env.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" >> $OUT/bench.err; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/bench_g2.json 2> $OUT/bench_g2.err; echo "exit $?" >> $OUT/bench_g2.err; cut -c1-300 $OUT/bench_g2.json; tail -2 $OUT/bench_g2.err
timeout 600 python bench.py --gpus 2 --config 5 --steps 2 --warmup 1 > $OUT/bench_g2_c5.json 2> $OUT/bench_g2_c5.err; echo "exit $?" >> $OUT/bench_g2_c5.err; cut -c1-300 $OUT/bench_g2_c5.json; tail -2 $OUT/bench_g2_c5.err
timeout 600 python bench.py --gpus 4 --steps 2 --warmup 1 > $OUT/bench_g4.json 2> $OUT/bench_g4.err; echo "exit $?" >> $OUT/bench_g4.err; cut -c1-300 $OUT/bench_g4.json; tail -2 $OUT/bench_g4.err

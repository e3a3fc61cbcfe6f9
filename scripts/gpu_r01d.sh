OUT=$PWD/gpurun_out/r01d; mkdir -p $OUT
timeout 300 ./tools/kernel_bench > $OUT/kernel_bench.txt 2>&1; cat $OUT/kernel_bench.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -15 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
bash scripts/gpu_profile.sh r01d/prof

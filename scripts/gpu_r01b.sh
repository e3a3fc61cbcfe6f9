OUT=$PWD/gpurun_out/r01b; mkdir -p $OUT
timeout 300 ./tools/ubench_f64 > $OUT/ubench_f64.txt 2>&1
timeout 300 ./tools/kernel_bench > $OUT/kernel_bench.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "synthetic or layout" > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
rocprofv3 -L > $OUT/counters.txt 2>&1
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --engine valu > $OUT/bench_valu.json 2> $OUT/bench_valu.err
cat $OUT/kernel_bench.txt; tail -3 $OUT/pytest_gpu.txt; tail -22 $OUT/ubench_f64.txt | head -16; cat $OUT/bench_valu.json

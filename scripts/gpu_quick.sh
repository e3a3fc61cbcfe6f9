# quick perf iteration: kernel bench + subset of tests + bench (no cpu leg)
TAG=${1:-q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$UBENCH" ]; then timeout 300 ./tools/ubench_f64 > $OUT/ubench_f64.txt 2>&1; grep "4x4x4" $OUT/ubench_f64.txt; fi
timeout 300 ./tools/kernel_bench 5120 65536 > $OUT/kernel_bench.txt 2>&1; cat $OUT/kernel_bench.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x ${PYTEST_K} > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('BENCH', round(d['value']), 'pts/s  exec TF', round(d['roofline']['executed_tflops'],2), d['phases_ms_per_step'])"; tail -2 $OUT/bench.err

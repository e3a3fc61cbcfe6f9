OUT=$PWD/gpurun_out/r01k; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -6 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
for c in 3 4 5; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_c$c.json')); print('C$c', round(d['value']), 'pts/s exec TF', round(d['roofline']['executed_tflops'],2), 'eff', round(d['roofline']['achieved'],1), d['phases_ms_per_step'], d['config']['factor_path'])"; done
bash scripts/gpu_profile.sh r01k/prof > /dev/null 2>&1
python scripts/pmc_summary.py $OUT/prof | grep "contract\|k_rhs\|k_cvec\|k_update"

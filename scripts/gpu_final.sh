# End-of-round evidence run: micro-benchmarks, kernel bench (with A/B and ablations), GPU tests, all bench configs, profiles
OUT=$PWD/gpurun_out/${1:-final}; mkdir -p $OUT
timeout 300 ./tools/ubench_f64 > $OUT/ubench_f64.txt 2>&1
timeout 300 ./tools/kernel_bench 5120 65536 > $OUT/kernel_bench.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
for c in 3 4 5; do timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; done
for k in 10 50; do timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --moving-window $k > $OUT/bench_mw$k.json 2>> $OUT/bench.err; done
timeout 300 python scripts/stat_time.py > $OUT/stat_time.txt 2>&1
bash scripts/gpu_profile.sh ${1:-final}/prof > /dev/null 2>&1
python scripts/pmc_summary.py $OUT/prof | grep "contract"
timeout 400 python scripts/mw_big_time.py > $OUT/mw_big_time.txt 2>&1
timeout 200 python scripts/execute_overhead.py 2>&1 | head -3 > $OUT/execute_overhead.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err
( REPO=$PWD; cd /tmp && export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mw -o mw -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --moving-window 50 > $OUT/prof_mw.json 2> $OUT/prof_mw.err )

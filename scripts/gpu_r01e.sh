OUT=$PWD/gpurun_out/r01e; mkdir -p $OUT
timeout 300 ./tools/kernel_bench 5120 65536 > $OUT/kernel_bench.txt 2>&1; cat $OUT/kernel_bench.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --chunk 65536 > $OUT/bench_c64k.json 2>> $OUT/bench.err; cat $OUT/bench_c64k.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do n=$(echo $c | cut -d_ -f1-2 | tr ' ' _); timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2> $OUT/pmc_$n.err; done
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $OUT | grep contract

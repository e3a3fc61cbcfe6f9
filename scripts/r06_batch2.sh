#!/bin/bash
# round 6, second GPU call: the suite after the prune + the triangle exchange; the ABL & 32 number (B tile generated on the VALU inside the K loop)
# at config 2's shape; the kernel trace of a config-5 prediction for its timeline; where execute() spends its time at k = 10 (moving window)
REPO=$PWD; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -s --durations=12 ) > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt
grep -E "passed|failed|^real|^FAILED" $OUT/pytest_gpu.txt | cut -c1-300
timeout 300 ./tools/kernel_bench 5120 32768 > $OUT/kernel_bench_c2.txt 2>&1; grep -E "ablate|k_contract<sym>" $OUT/kernel_bench_c2.txt | cut -c1-200
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c5 -o t -- python $REPO/bench.py --config 5 --steps 2 --warmup 1 --no-cpu --pmc off --no-other > $OUT/trace_c5.json 2> $OUT/trace_c5.err
cd $REPO
python scripts/predict_timeline.py $OUT/trace_c5 full > $OUT/predict_timeline_c5.txt 2>&1; head -40 $OUT/predict_timeline_c5.txt | cut -c1-200
rm -rf $OUT/trace_c5/*/*.db 2>/dev/null; find $OUT/trace_c5 -name "*.csv" -size +20M -delete
timeout 300 python scripts/execute_breakdown.py 2 --window 10 > $OUT/execute_breakdown_mw10.txt 2>&1; tail -30 $OUT/execute_breakdown_mw10.txt | cut -c1-200

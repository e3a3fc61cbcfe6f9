#!/bin/bash
TAG=${1:-r2j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; grep -n "FAILED\|Error" $OUT/pytest_gpu.txt | head
for c in 2 3 4 5; do timeout 900 python bench.py --steps 3 --warmup 1 --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; python -c "
import json; d=json.load(open('$OUT/bench_c$c.json')); cb=d.get('cpu_baseline',{}); print('BENCH c$c', round(d['value']), 'pts/s', round(d['roofline']['achieved'],2), d['roofline']['traffic'], d['phases_ms_per_step'], 'pcie', round(d['pcie_inclusive']['value']), 'cpu', cb.get('kind'), cb.get('value'), cb.get('steady_state'), cb.get('cores'))"; done
timeout 300 python scripts/small_problem_latency.py > $OUT/small_problem_latency.txt 2>&1; cat $OUT/small_problem_latency.txt
timeout 300 python scripts/inverse_lookahead_ab.py > $OUT/inverse_ab.txt 2>&1; cat $OUT/inverse_ab.txt

#!/bin/bash
# round 3, second session: A/B of the triangular diagonal blocks of the contraction (option tri) and of the sweep's panel stream
# (option panel_stream), then the parity tests that exercise both.
TAG=${1:-tp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 200 ./tools/kernel_bench 5120 65536 5120 tri > $OUT/kernel_bench_tri.txt 2>&1; cat $OUT/kernel_bench_tri.txt
timeout 300 python scripts/sweep_option_ab.py panel_stream 0 1 --configs=3,4,2,5 > $OUT/panel_stream_ab.txt 2>&1; cat $OUT/panel_stream_ab.txt
for t in 0 1; do MIK_TRI=$t timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off > $OUT/bench_tri$t.json 2> $OUT/bench_tri$t.err; python - <<P
import json
d=json.load(open('$OUT/bench_tri$t.json')); r=d['roofline']
print('MIK_TRI=$t', round(d['value']), 'pts/s; contract ms', round(d['phases_ms_per_step']['contract'],2), 'invert', round(d['phases_ms_per_step']['invert'],2), 'executed TF', round(r['achieved'],2), 'frac', round(r['frac'],4), 'checksum', d['checksum']['z_sum'], d['checksum']['ss_sum'])
P
done
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt

#!/bin/bash
# round 6, third GPU call: the exchange with mirrored triangles, the parallel k_sp_tiles_g + folded memsets (config 5 timeline again), the k = 10 bench line
REPO=$PWD; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_device_group.py tests/test_sparse_contraction.py tests/test_exchange_deadline.py tests/test_fullsize_and_host_rules.py -m gpu -q --tb=short -x ) > $OUT/pytest_subset.txt 2>&1
grep -E "passed|failed|^real|^FAILED|^E  " $OUT/pytest_subset.txt | cut -c1-300 | head -20
timeout 400 python bench.py --steps 5 --warmup 2 --config 5 --no-other --no-cpu --pmc off > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - $OUT/bench_c5.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config 5: %.4g points/s, %.2f ms/step, phases %s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["phases_ms_per_step"].items()}))
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c5 -o t -- python $REPO/bench.py --config 5 --steps 2 --warmup 1 --no-cpu --pmc off --no-other > $OUT/trace_c5.json 2> $OUT/trace_c5.err
cd $REPO
python scripts/predict_timeline.py $OUT/trace_c5 full > $OUT/predict_timeline_c5.txt 2>&1; head -22 $OUT/predict_timeline_c5.txt | cut -c1-200; sed -n 60,100p $OUT/predict_timeline_c5.txt
rm -rf $OUT/trace_c5
python - <<'PY'
import bench, os
os.environ["MIK_FACTOR_CACHE"] = "0"
for steps, warm in ((3, 1), (20, 3), (3, 1)):
    for k in (10, 100):
        l = bench.other_config_line(2, k, steps=steps, warmup=warm)
        print("moving window k=%d, steps %d warmup %d: %.3f ms per step, %.4g points/s; phases %s" % (k, steps, warm, l["ms_per_step"], l["value"], l["phases_ms_per_step"]))
PY

#!/bin/bash
# round 6, fourth GPU call: exchange tests again, config 5 with serialized contractions, the moving-window matrix-core gate, the bench's k = 10 line after the pool fix
REPO=$PWD; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_device_group.py tests/test_sparse_contraction.py tests/test_exchange_deadline.py -m gpu -q --tb=short ) > $OUT/pytest_subset.txt 2>&1
grep -E "passed|failed|^real|^FAILED|^E  " $OUT/pytest_subset.txt | cut -c1-400 | head -20
timeout 300 ./tools/mw_ldl_bench 200000 > $OUT/mw_ldl_bench.txt 2>&1; cat $OUT/mw_ldl_bench.txt | cut -c1-330
for i in 1 2; do
timeout 400 python bench.py --steps 5 --warmup 2 --config 5 --no-other --no-cpu --pmc off > $OUT/bench_c5_$i.json 2> $OUT/bench_c5.err
python - $OUT/bench_c5_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("config 5: %.4g points/s, %.2f ms/step, phases %s" % (d["value"], d["ms_per_step"], {k: round(v, 2) for k, v in d["phases_ms_per_step"].items()}))
PY
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c5 -o t -- python $REPO/bench.py --config 5 --steps 2 --warmup 1 --no-cpu --pmc off --no-other > $OUT/trace_c5.json 2> $OUT/trace_c5.err
cd $REPO
python scripts/predict_timeline.py $OUT/trace_c5 full > $OUT/predict_timeline_c5.txt 2>&1; head -22 $OUT/predict_timeline_c5.txt | cut -c1-200; sed -n 60,90p $OUT/predict_timeline_c5.txt
rm -rf $OUT/trace_c5
( time timeout 900 python bench.py --no-cpu --pmc off ) > $OUT/bench_c2_nocpu.json 2> $OUT/bench_c2_nocpu.err
python - $OUT/bench_c2_nocpu.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("headline %.4g points/s (%.2f ms/step, frac %.3f); c3 %.4g c4 %.4g c5 %.4g (%.2f ms) mw10 %.4g (%.3f ms) mw100 %.4g" % (d["value"], d["ms_per_step"], d["roofline"]["frac"],
      c["c3_value"], c["c4_value"], c["c5_value"], c["c5_ms_per_step"], c["mw_k10_value"], c["mw_k10_ms_per_step"], c["mw_k100_value"]))
PY

#!/bin/bash
REPO=$PWD; export TMPDIR=/tmp
( time timeout 1200 python bench.py --full-parity 2,4,3,5 ) > $OUT/full_grid_parity.jsonl 2> $OUT/full_grid_parity.err; echo "exit $?" >> $OUT/full_grid_parity.err; cut -c1-500 $OUT/full_grid_parity.jsonl; tail -4 $OUT/full_grid_parity.err
( time timeout 900 python -m pytest tests/test_zz_full_grid_parity.py -m gpu -q -s --tb=short ) > $OUT/pytest_fullgrid.txt 2>&1; grep -E "^config|passed|failed|^real" $OUT/pytest_fullgrid.txt | cut -c1-420
( time timeout 1200 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python - $OUT/bench_c2.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print("headline %.4g points/s (%.2f ms/step, frac %.3f, launch %.2f ms); c3 %.4g c4 %.4g c5 %.4g (%.2f ms) mw10 %.4g (%.3f ms) mw100 %.4g; cpu %.1f (%d threads); full grid %d points dz %.2e dss %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], c["roofline_avg_launch_ms"],
      c["c3_value"], c["c4_value"], c["c5_value"], c["c5_ms_per_step"], c["mw_k10_value"], c["mw_k10_ms_per_step"], c["mw_k100_value"], c["cpu_value"], c["cpu_cores"], c["c2_fullgrid_points_checked"], c["c2_fullgrid_max_abs_dz"], c["c2_fullgrid_max_abs_dss"]))
PY
tail -3 $OUT/bench_c2.err

#!/bin/bash
# round 4, second session, GPU call 13: style='points' without host copies of the coordinates -- the whole GPU suite, points-style rates
OUT=$PWD/gpurun_out/${1:-r04b_13}; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
for c in 5 2 3; do timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --no-other --config $c > $OUT/bench_c$c.json 2> $OUT/bench.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c$c.json").read().strip().split("\n")[-1])
print("config $c grid", round(d["value"]), "points", round(d["execute_points_style"]["value"]), "resident", round(d["resident"]["value"]), d["checksum"]["grid_vs_points_max_abs_dz"], d["checksum"]["grid_vs_points_max_abs_dss"])
EOF
done | tee $OUT/points_style.txt

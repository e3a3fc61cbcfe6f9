#!/bin/bash
# round 4, second session, GPU call 9: group size of k_contract_spg's queue order (sparse_group)
OUT=$PWD/gpurun_out/${1:-r04b_9}; mkdir -p $OUT
for g in 1 16; do MIK_SPARSE_GROUP=$g timeout 300 python -m pytest tests/test_sparse_contraction.py -m gpu -x -q --tb=short 2>&1 | tail -1; done > $OUT/pytest_sparse.txt 2>&1; cat $OUT/pytest_sparse.txt
for g in 4 1 2 8 16 4; do MIK_SPARSE_GROUP=$g timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --pmc off --no-other --config 5 > $OUT/bench_c5_g$g.json 2> $OUT/bench_c5.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c5_g$g.json").read().strip().split("\n")[-1])
print("sparse_group $g", round(d["value"]), round(d["ms_per_step"],2), round(d["phases_ms_per_step"]["contract"],2), round(d["roofline"]["achieved"],2))
EOF
done | tee $OUT/sparse_group_ab.txt

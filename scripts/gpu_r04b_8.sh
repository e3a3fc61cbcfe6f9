#!/bin/bash
# round 4, second session, GPU call 8: k_contract_spg's epilogue from the B tile in LDS (sparse_epilogue 1) against global memory (0)
OUT=$PWD/gpurun_out/${1:-r04b_8}; mkdir -p $OUT
for ep in 1 0; do MIK_SPARSE_EPILOGUE=$ep timeout 300 python -m pytest tests/test_sparse_contraction.py -m gpu -x -q --tb=short 2>&1 | tail -2; done > $OUT/pytest_sparse.txt 2>&1; cat $OUT/pytest_sparse.txt
for ep in 0 1 0 1; do MIK_SPARSE_EPILOGUE=$ep timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu --pmc off --no-other --config 5 > $OUT/bench_c5_epi$ep.json 2> $OUT/bench_c5.err; python - <<EOF
import json
d=json.loads(open("$OUT/bench_c5_epi$ep.json").read().strip().split("\n")[-1])
print("sparse_epilogue $ep", round(d["value"]), round(d["ms_per_step"],2), d["phases_ms_per_step"]["contract"], d["roofline"]["achieved"])
EOF
done
MIK_FUZZ_CASES=300 timeout 300 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -3

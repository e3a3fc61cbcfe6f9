#!/bin/bash
run() { MIK_SPG_RESERVE=$2 timeout 300 python bench.py --steps 6 --warmup 2 --config 5 --no-other --no-cpu --pmc off --sparse-lanes $1 > $OUT/c5_L$1_R$2.json 2> $OUT/c5.err
  python - $OUT/c5_L$1_R$2.json $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["phases_ms_per_step"]
print("lanes %s reserve %3s: %.4g points/s, %.2f ms/step, contract %.2f rhs %.2f predict_total %.2f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], p["contract"], p["rhs"], p["predict_total"]))
PY
}
run 2 0; run 3 0; run 3 16; run 3 32; run 3 48; run 3 64; run 3 96; run 2 0; run 3 32
timeout 600 python -m pytest tests/test_sparse_contraction.py -m gpu -q -x 2>&1 | tail -2

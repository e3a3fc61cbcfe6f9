# chunk size x rhs_overlap A/B of the predict loop (config 2 by default): ms/step, contraction TFLOP/s, rhs ms
CFG=${1:-2}; OUT=gpurun_out/${2:-r03c}; mkdir -p $OUT
for ov in 0 1; do for ch in 65536 131072 262144 524288 1048576; do
  MIK_RHS_OVERLAP=$ov timeout 300 python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu --pmc off --chunk $ch > $OUT/sweep.json 2> $OUT/sweep.err || { tail -3 $OUT/sweep.err; continue; }
  python - <<PY
import json; d=json.load(open("$OUT/sweep.json")); p=d["phases_ms_per_step"]; r=d["roofline"]
print("cfg $CFG overlap $ov chunk %8d : %9.0f pts/s  %.2f ms/step  resident %.2f ms  contract %.2f ms (%.2f TF, %d launches)  rhs %.2f  predict %.2f  host %.2f ms" % ($ch, d["value"], d["ms_per_step"], d["resident"]["ms_per_step"], p["contract"], r["achieved"], r["launches_per_step"], p["rhs"], p["predict_total"], d["host_overhead"]["ms_per_step"]))
PY
done; done 2>&1 | tee -a $OUT/chunk_sweep_c$CFG.txt

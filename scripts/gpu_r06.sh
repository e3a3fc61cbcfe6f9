#!/bin/bash
# Round-6 GPU runs, one parameterised script.   bash scripts/gpu_r06.sh STAGE [TAG]   -> output under gpurun_out/r06_TAG/
# STAGES
#   tests       the whole GPU suite with per-test durations (incl. the four oracle cases beyond N = 8000 and the whole-grid parity tests)
#   fullparity  bench.py --full-parity: configs 2, 4, 3 whole grids + the config-5 strip against the staged reference, unbounded
#   bench       the default bench run (config 2 + other configs + CPU leg + bounded whole-grid parity + live PMC)
#   first       env + tests + fullparity + bench (the round's first contact)
#   evidence    end-of-round: env, tests, bench, kernel stats + PMC passes of configs 2 and 5, aliased 8-member group at config 5
#   timeline    kernel trace of a config-5 prediction -> scripts/predict_timeline.py (profiles/r06_predict_timeline_c5_*.txt)
#   gates       tools/mw_ldl_bench (moving-window elimination on the matrix cores), tools/kernel_bench (B tile on the VALU), scripts/r06_diag_mw10.py
# (the round's one-shot batches -- suite after the prune, exchange tests, A/Bs of the range-aware prediction -- ran through this script as
#  scripts/r06_<stage>.sh files that are gone again; their outputs are under profiles/r06_*)
STAGE=${1:-tests}; TAG=${2:-$STAGE}; OUT=$PWD/gpurun_out/r06_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
env_probe() { { nproc; free -g | head -2; grep -m1 "model name" /proc/cpuinfo; /opt/rocm/bin/rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9|Compute Unit"; } > $OUT/env.txt 2>&1; }
tests() { ( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -s --durations=25 ) > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "^config [0-9]|^n[0-9]+_|passed|failed|^real" $OUT/pytest_gpu.txt | cut -c1-400; }
fullparity() { ( time timeout 2400 python bench.py --full-parity ${FULLPARITY_CONFIGS:-2,4,3,5} ) > $OUT/full_grid_parity.jsonl 2> $OUT/full_grid_parity.err; echo "exit $?" >> $OUT/full_grid_parity.err; cut -c1-700 $OUT/full_grid_parity.jsonl; tail -4 $OUT/full_grid_parity.err; }
bench() { ( time timeout 1200 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; python - $OUT/bench_c2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value %.4g  ms/step %.2f  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
    print(json.dumps(d["config"])[:3000])
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -4 $OUT/bench_c2.err; }
profile() {  # kernel stats + PMC passes of config $1
  local c=$1
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c$c -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --no-other --config $c > $OUT/ks_c$c.json 2> $OUT/ks_c$c.err
  run() { local name=$1; shift; timeout 400 rocprofv3 "$@" --output-format csv -d $OUT/prof$c/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other --config $c > $OUT/prof${c}_$name.json 2> $OUT/prof${c}_$name.err; }
  run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
  run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum
  run pmc_fetch --kernel-trace --pmc FETCH_SIZE
  run pmc_write --kernel-trace --pmc WRITE_SIZE
  python $REPO/scripts/pmc_summary.py $OUT/prof$c > $OUT/pmc_per_kernel_c$c.csv
  cd $REPO
  rm -rf $OUT/prof*/*/*.db $OUT/ks_c*/*.db 2>/dev/null
}
case $STAGE in
tests) tests;;
fullparity) fullparity;;
bench) bench;;
first) env_probe; cat $OUT/env.txt; tests; fullparity; bench;;
evidence)
  env_probe; tests; bench
  timeout 500 python bench.py --steps 3 --warmup 1 --config 5 --no-other > $OUT/bench_c5.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_c5.json
  profile 2; profile 5
  timeout 400 python bench.py --gpus 8 --config 5 --steps 2 --warmup 1 --no-cpu > $OUT/bench_g8_c5.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_g8_c5.json
  ;;
timeline)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c5 -o t -- python $REPO/bench.py --config 5 --steps 2 --warmup 1 --no-cpu --pmc off --no-other > $OUT/trace_c5.json 2> $OUT/trace_c5.err
  cd $REPO
  python scripts/predict_timeline.py $OUT/trace_c5 full > $OUT/predict_timeline_c5.txt 2>&1; head -24 $OUT/predict_timeline_c5.txt | cut -c1-200
  rm -rf $OUT/trace_c5
  ;;
gates)
  timeout 300 ./tools/mw_ldl_bench 200000 > $OUT/mw_ldl_bench.txt 2>&1; cat $OUT/mw_ldl_bench.txt | cut -c1-300
  timeout 300 ./tools/kernel_bench 5120 32768 > $OUT/kernel_bench_c2.txt 2>&1; grep -E "ablate 8-wave" $OUT/kernel_bench_c2.txt
  timeout 900 python scripts/r06_diag_mw10.py > $OUT/diag_mw10.txt 2>&1; cut -c1-300 $OUT/diag_mw10.txt
  ;;
*) if [ -f scripts/r06_$STAGE.sh ]; then OUT=$OUT bash scripts/r06_$STAGE.sh; else echo "unknown stage $STAGE"; exit 2; fi;;
esac

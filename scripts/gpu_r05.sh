#!/bin/bash
# Round-5 GPU runs, one parameterised script (the review asked for one instead of sixteen one-shots).
#   bash scripts/gpu_r05.sh STAGE [TAG]      -> output under gpurun_out/r05_TAG/
# STAGES
#   k2      the sweep tests + A/B of the deep trailing update (update_deep, update_tpb) at bench configs 3, 4, 2, 5
#   largen  parity beyond N = 8000 against the oracle (tests/test_large_n.py with MIK_SLOW_TESTS=1)
#   tests   the whole GPU suite
#   bench   bench.py at config 2 (default run, incl. other_configs + CPU leg + live PMC)
#   evidence  tests + bench + rocprofv3 kernel stats of the bench (profiles for the round)
STAGE=${1:-k2}; TAG=${2:-$STAGE}; OUT=$PWD/gpurun_out/r05_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
case $STAGE in
k2)
  timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "sweep or panel_stream or inverse" > $OUT/pytest_k2.txt 2>&1; tail -3 $OUT/pytest_k2.txt
  for opt in "update_deep 0 1" "update_tpb 1 2 3 4 6 8 16"; do
    timeout 900 python scripts/sweep_option_ab.py $opt 2>&1 | tee -a $OUT/update_deep_ab.txt
  done
  ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
  ;;
largen)
  MIK_SLOW_TESTS=1 timeout 1500 python -m pytest tests/test_large_n.py -m gpu -q -s > $OUT/pytest_large_n.txt 2>&1; grep -E "^n[0-9]|passed|failed|Error" $OUT/pytest_large_n.txt
  ;;
bench)
  timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 1500 $OUT/bench_c2.json
  ;;
evidence)
  # end-of-round evidence in one call: environment, the whole GPU suite, the default bench (config 2 + other_configs + CPU leg + live PMC),
  # kernel stats and PMC passes of the same command, config 5 with kernel stats, tools/update_bench, parity beyond N = 8000, randomized campaign,
  # the 8-member aliased group at config 5
  REPO=$PWD
  { nproc; free -g | head -2; grep -m1 "model name" /proc/cpuinfo; /opt/rocm/bin/rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9|Compute Unit"; } > $OUT/env.txt 2>&1
  timeout 1500 python -m pytest tests -m gpu -q --tb=short > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
  ( time timeout 900 python bench.py ) > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-160 $OUT/bench_c2.json; tail -3 $OUT/bench_c2.err
  timeout 500 python bench.py --steps 3 --warmup 1 --config 5 --no-other > $OUT/bench_c5.json 2>> $OUT/bench.err; cut -c1-160 $OUT/bench_c5.json
  [ -n "$EVIDENCE_LITE" ] || timeout 300 ./tools/update_bench 8064 1 > $OUT/update_bench.txt 2>&1
  cd /tmp
  CFGS="2 5"; [ -n "$EVIDENCE_LITE" ] && CFGS="5"   # EVIDENCE_LITE=1: only what changed after the full run (config 5's kernels)
  for c in $CFGS; do
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c$c -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --no-other --config $c > $OUT/ks_c$c.json 2> $OUT/ks_c$c.err
    run() { local name=$1; shift; timeout 400 rocprofv3 "$@" --output-format csv -d $OUT/prof$c/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other --config $c > $OUT/prof${c}_$name.json 2> $OUT/prof${c}_$name.err; }
    run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
    run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum
    run pmc_fetch --kernel-trace --pmc FETCH_SIZE
    run pmc_write --kernel-trace --pmc WRITE_SIZE
    python $REPO/scripts/pmc_summary.py $OUT/prof$c > $OUT/pmc_per_kernel_c$c.csv
  done
  cd $REPO
  MIK_SLOW_TESTS=1 timeout 1200 python -m pytest tests/test_large_n.py -m gpu -q -s > $OUT/pytest_large_n.txt 2>&1; grep -E "^n[0-9]|passed|failed" $OUT/pytest_large_n.txt
  [ -n "$EVIDENCE_LITE" ] || { echo "MIK_FUZZ_CASES=2000 python -m pytest tests/test_randomized_parity.py -m gpu -q -s   (MI355X, HEAD of round 5)"; MIK_FUZZ_CASES=2000 timeout 900 python -m pytest tests/test_randomized_parity.py -m gpu -q -s 2>&1 | tail -4; } > $OUT/randomized.txt 2>&1; tail -3 $OUT/randomized.txt
  timeout 400 python bench.py --gpus 8 --config 5 --steps 2 --warmup 1 --no-cpu > $OUT/bench_g8_c5.json 2>> $OUT/bench.err; cut -c1-120 $OUT/bench_g8_c5.json
  [ -n "$EVIDENCE_LITE" ] || timeout 300 python scripts/mw_static_ab.py > $OUT/mw_static_ab.txt 2>&1; grep -E "k=100|k= 50" $OUT/mw_static_ab.txt | head -4
  rm -rf $OUT/prof*/*/*.db $OUT/ks_c*/*.db 2>/dev/null
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac

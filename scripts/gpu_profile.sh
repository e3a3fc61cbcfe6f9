#!/bin/bash
# rocprofv3 passes over bench.py (config 2, 1 timed step): kernel trace + stats, then PMC passes (each alone).
# Usage: bash scripts/gpu_profile.sh TAG [extra bench args]
TAG=${1:-prof}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
run() { # name, rocprof args...
  local name=$1; shift
  timeout 600 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu "${BENCH_EXTRA[@]}" > $OUT/$name.json 2> $OUT/$name.err
}
BENCH_EXTRA=("$@")
run ktrace --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64
run pmc_lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU TCC_HIT_sum TCC_MISS_sum
cd $REPO
find $OUT -name "*.csv" | head -30

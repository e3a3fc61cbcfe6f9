OUT=$PWD/gpurun_out/${1:-r01i}; mkdir -p $OUT
./tools/kernel_bench 5120 65536 2>&1 | grep "contract\|update"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/tcc -o p -- $GRAFT_REPO_ROOT/tools/kernel_bench 5120 65536 > $OUT/kb_tcc.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $GRAFT_REPO_ROOT/tools/kernel_bench 5120 65536 > $OUT/kb_fetch.txt 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $OUT | grep "k_contract<"

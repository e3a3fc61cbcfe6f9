OUT=$PWD/gpurun_out/r01h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/sq -o p -- $GRAFT_REPO_ROOT/tools/kernel_bench 5120 65536 > $OUT/kb_sq.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU --output-format csv -d $OUT/tcc -o p -- $GRAFT_REPO_ROOT/tools/kernel_bench 5120 65536 > $OUT/kb_tcc.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $GRAFT_REPO_ROOT/tools/kernel_bench 5120 65536 > $OUT/kb_fetch.txt 2>&1
cd $GRAFT_REPO_ROOT; python scripts/pmc_summary.py $OUT | grep "k_contract<"

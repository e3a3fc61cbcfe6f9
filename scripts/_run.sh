cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r05_pmc8; mkdir -p $OUT
for kt in 16 8; do
  run() { local name=$1; shift; MIK_SPARSE_KTILE=$kt timeout 300 rocprofv3 "$@" --output-format csv -d $OUT/k$kt/$name -o $name -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other --config 5 > $OUT/k${kt}_$name.json 2> $OUT/k${kt}_$name.err; }
  run pmc_sq --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY
  run pmc_tcc --kernel-trace --pmc SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum
  run pmc_fetch --kernel-trace --pmc FETCH_SIZE
  python $R/scripts/pmc_summary.py $OUT/k$kt > $OUT/pmc_per_kernel_k$kt.csv
  grep "k_contract_spg\|Kernel" $OUT/pmc_per_kernel_k$kt.csv | cut -c1-400
  rm -rf $OUT/k$kt/*/*.db
done

# K3b tile-shape experiment (round 3): 128x128 (library) vs 256x128 block tile of the symmetric contraction, times from
# tools/kernel_bench and per-kernel PMC counters (fabric traffic, L2 hit rate, MFMA-busy, clock) from separate rocprofv3 passes
OUT=$PWD/gpurun_out/${1:-r03tile}; mkdir -p $OUT; REPO=$PWD
timeout 600 ./tools/kernel_bench 5120 65536 > $OUT/kernel_bench.txt 2>&1; grep -E "TILE|256x128|k_contract<sym>  mfma 8" $OUT/kernel_bench.txt
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/tile_pmc/$name -o $name -- $REPO/tools/kernel_bench 5120 65536 > /dev/null 2> $OUT/tile_pmc_$name.err
done
cd $REPO
python scripts/pmc_summary.py $OUT/tile_pmc > $OUT/tile_pmc_per_kernel.csv
grep -E "k_contract<true, 2, true, false>|k_contract256<true>|k_contract<false, 2, true, false>|k_contract256<false>" $OUT/tile_pmc_per_kernel.csv

#!/bin/bash
# Gate 2 of the GEMM-rich K2 form: the sweep at M/2 and M, and the tile engine's product rates at M/2 (tools/kernel_bench Mp P = M/2 M/2)
OUT=$PWD/gpurun_out/r06_${2:-k2gate}; mkdir -p $OUT
timeout 600 python scripts/k2_recursive_gate.py > $OUT/sweep.txt 2>&1; cat $OUT/sweep.txt
for s in 2048 2560 4096; do echo "== tools/kernel_bench $s $s"; timeout 300 tools/kernel_bench $s $s 2>&1 | grep -E "A/B full 8-wave|A/B sym 8-wave (one|persistent \()|k_contract<(full|sym)>  ?mfma 8|k_update" ; done > $OUT/products.txt 2>&1; cat $OUT/products.txt

#!/bin/bash
# round 4, second session, GPU call 1: parity of the gathered-row-group contraction, A/B against the aligned form, kernel trace of config 5
OUT=$PWD/gpurun_out/${1:-r04b_1}; mkdir -p $OUT; REPO=$PWD
timeout 400 python -m pytest tests/test_sparse_contraction.py -m gpu -x -q --tb=short > $OUT/pytest_sparse.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_sparse.txt; tail -15 $OUT/pytest_sparse.txt
timeout 400 python scripts/sparse_rows_ab.py --dense > $OUT/sparse_rows_ab.txt 2>&1; echo "exit $?" >> $OUT/sparse_rows_ab.txt; cat $OUT/sparse_rows_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c5 -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config 5 > $OUT/ks_c5.json 2> $OUT/ks_c5.err
cd $REPO; head -14 $OUT/ks_c5/*kernel_stats.csv | cut -c1-150; cut -c1-300 $OUT/ks_c5.json
rm -rf $OUT/ks_c5/*.db $OUT/ks_c5/*/*.db 2>/dev/null

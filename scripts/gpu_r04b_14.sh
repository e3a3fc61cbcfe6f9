#!/bin/bash
# round 4, second session, GPU call 14: kernel trace of the moving window (k = 10) over sorted shuffled points against the rows of a grid
OUT=$PWD/gpurun_out/${1:-r04b_14}; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python $REPO/scripts/mw_sorted_trace.py > $OUT/run.txt 2>&1
cd $REPO; cat $OUT/run.txt | tail -3; cut -d, -f1-4 $OUT/ks/ks_kernel_stats.csv | sed 's/(.*)"/"/' | head -24
rm -f $OUT/ks/*.db

#!/bin/bash
TAG=${1:-r2e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "symsweep=0 fused_panel=0" "symsweep=0 fused_panel=1" "symsweep=1"; do
  n=$(echo $v | tr ' =' '__')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$n -o tl -- python $REPO/scripts/inverse_timeline.py run 5000 $v > $OUT/tl_$n.txt 2>&1
  (cd $REPO; echo "== N=5000 $v"; grep invert_ms $OUT/tl_$n.txt; python scripts/inverse_timeline.py parse $OUT/tl_$n) | tee -a $OUT/inverse_timeline.txt
done
for v in "symsweep=0" "symsweep=1"; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_8000_$v -o tl -- python $REPO/scripts/inverse_timeline.py run 8000 $v > $OUT/tl_8000_$v.txt 2>&1
(cd $REPO; echo "== N=8000 $v"; grep invert_ms $OUT/tl_8000_$v.txt; python scripts/inverse_timeline.py parse $OUT/tl_8000_$v) | tee -a $OUT/inverse_timeline.txt
done

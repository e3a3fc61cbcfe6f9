#!/bin/bash
# kernel timeline of one inverse per size / option set: scripts/gpu_timeline.sh <tag> "<N> [opt=val ...]" ...
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$i -o tl -- python $REPO/scripts/inverse_timeline.py run $spec > $OUT/tl_$i.txt 2>&1
  { echo "== $spec"; grep invert_ms $OUT/tl_$i.txt; python $REPO/scripts/inverse_timeline.py parse $OUT/tl_$i; } >> $OUT/timeline.txt 2>&1
done
cd $REPO; cat $OUT/timeline.txt | cut -c1-150

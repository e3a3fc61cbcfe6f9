#!/bin/bash
TAG=${1:-k2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -x -k "sweep or inverse or factor" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for n in 5000 8000; do
for v in $VARIANTS; do
  t=$(echo ${n}_$v | tr ',=' '__')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$t -o tl -- python $REPO/scripts/inverse_timeline.py run $n $(echo $v | tr ',' ' ') > $OUT/tl_$t.txt 2>&1
  (cd $REPO; echo "== N=$n $v"; grep invert_ms $OUT/tl_$t.txt; python scripts/inverse_timeline.py parse $OUT/tl_$t 2>&1 | grep -E "one factor|k_diag_inv|k_update|k_panel|k_gate|step period|large") | tee -a $OUT/inverse_timeline.txt
done; done

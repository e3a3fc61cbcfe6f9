#!/bin/bash
# K2: fused panel + half sweep; timeline of one inverse; parity; bench configs 2-5 with CPU legs; kernel stats for each config.
TAG=${1:-r2d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt; grep -n "FAILED\|Error" $OUT/pytest_gpu.txt | head
grep -n "\] N=" $OUT/pytest_gpu.txt | cut -c1-170
cd /tmp && export TMPDIR=/tmp
for v in "symsweep=0 fused_panel=0" "symsweep=0 fused_panel=1" "symsweep=1"; do
  n=$(echo $v | tr ' =' '__')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$n -o tl -- python $REPO/scripts/inverse_timeline.py run 5000 $v > $OUT/tl_$n.txt 2>&1
  (cd $REPO; echo "== N=5000 $v"; tail -1 $OUT/tl_$n.txt; python scripts/inverse_timeline.py parse $OUT/tl_$n) | tee -a $OUT/inverse_timeline.txt
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_8000 -o tl -- python $REPO/scripts/inverse_timeline.py run 8000 symsweep=0 > $OUT/tl_8000.txt 2>&1
(cd $REPO; echo "== N=8000 symsweep=0"; tail -1 $OUT/tl_8000.txt; python scripts/inverse_timeline.py parse $OUT/tl_8000) | tee -a $OUT/inverse_timeline.txt
cd $REPO
for c in 2 3 4 5; do timeout 900 python bench.py --steps 3 --warmup 1 --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; python -c "
import json; d=json.load(open('$OUT/bench_c$c.json')); cb=d.get('cpu_baseline',{}); print('BENCH c$c', round(d['value']), 'pts/s', round(d['roofline']['achieved'],2), d['roofline']['traffic'], d['phases_ms_per_step'], 'pcie', round(d['pcie_inclusive']['value']), 'cpu', cb.get('kind'), cb.get('value'), cb.get('steady_state'), cb.get('cores'), cb.get('gpu_vs_cpu_max_abs_dz'), cb.get('gpu_vs_cpu_max_abs_dss'))"; done
cd /tmp
for c in 3 4 5; do timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c$c -o ks -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu --pmc off --config $c > $OUT/ks_c$c.json 2> $OUT/ks_c$c.err; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_c2 -o ks -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu --pmc off > $OUT/ks_c2.json 2> $OUT/ks_c2.err
cd $REPO; ls $OUT/ks_c2

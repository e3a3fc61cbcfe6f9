#!/bin/bash
# Moving-window LDL^T kernel: parity (all GPU tests), timing table, bench lines, reference benchmark shapes; half-sweep
# accuracy campaign; sustained clock of the contraction forms.
TAG=${1:-r2c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
timeout 1200 python -m pytest tests -m gpu -q --tb=short -s > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -4 $OUT/pytest_gpu.txt; grep -n "FAILED\|Error" $OUT/pytest_gpu.txt | head -20
grep -n "\[half_sweep\]\|\[default\]\|\[pivoted\]\|randomized parity:\|needed the widened\|moving-window " $OUT/pytest_gpu.txt
timeout 900 python scripts/mw_big_time.py > $OUT/mw_big_time.txt 2>&1; cat $OUT/mw_big_time.txt
timeout 600 python scripts/reference_benchmark_shapes.py > $OUT/reference_benchmark_shapes.txt 2>&1; cat $OUT/reference_benchmark_shapes.txt
for k in 10 50 100; do timeout 600 python bench.py --steps 3 --warmup 1 --moving-window $k > $OUT/bench_mw$k.json 2> $OUT/bench_mw$k.err; python -c "
import json; d=json.load(open('$OUT/bench_mw$k.json')); print('BENCH mw$k', round(d['value']), 'pts/s', d['roofline']['achieved'], d['roofline']['traffic'], d['phases_ms_per_step'], d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('gpu_vs_cpu_max_abs_dz'))"; done
timeout 900 python scripts/sweep_vs_pivot_accuracy.py 400 > $OUT/sweep_accuracy.txt 2>&1; cat $OUT/sweep_accuracy.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_clk -o clk -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/pmc_clk.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/ktrace.txt 2>&1
cd $REPO
python - <<PY > $OUT/pmc_clk_summary.txt
import csv, glob
dur={}
for f in glob.glob("$OUT/ktrace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_contract" in r["Kernel_Name"]: dur.setdefault(r["Kernel_Name"][18:52],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-6)
for f in glob.glob("$OUT/pmc_clk/**/*counter_collection.csv", recursive=True):
    by={}
    for r in csv.DictReader(open(f)):
        if "k_contract" in r["Kernel_Name"]:
            d=by.setdefault(r["Dispatch_Id"],{"k":r["Kernel_Name"][18:52]}); d[r["Counter_Name"]]=float(r["Counter_Value"])
    for d in sorted(by,key=int): print(d, by[d])
print({k:(min(v), sum(v)/len(v)) for k,v in dur.items()})
PY
cat $OUT/pmc_clk_summary.txt

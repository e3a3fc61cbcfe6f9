#!/bin/bash
# K3b pair-unit scheduling: A/B in the tools, PMC (L2 hits, fabric reads) of the three forms, parity subset, bench.
TAG=${1:-r2b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
for mp in 5120 8064 2048 4096; do timeout 300 ./tools/contract_ld_bench $mp 65536 > $OUT/ld_$mp.txt 2>&1; cat $OUT/ld_$mp.txt; done
timeout 300 ./tools/kernel_bench 5120 65536 > $OUT/kernel_bench.txt 2>&1; head -16 $OUT/kernel_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o tcc -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/pmc_tcc.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/pmc_fetch.txt 2>&1
cd $REPO
python - <<PY > $OUT/pmc_summary.txt
import csv, glob
for name in ("tcc","fetch"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if "k_contract" in r["Kernel_Name"]]
        by={}
        for r in rows: by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"]); by[r["Dispatch_Id"]]["k"]=r["Kernel_Name"][18:52]
        for d in sorted(by,key=int): print(name, d, by[d])
PY
cat $OUT/pmc_summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('BENCH', round(d['value']), 'pts/s', d['roofline']['achieved'], d['roofline']['traffic'], d['phases_ms_per_step'])"
MIK_PAIRS=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --pmc off > $OUT/bench_nopairs.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_nopairs.json')); print('BENCH nopairs', round(d['value']), 'pts/s', d['roofline']['achieved'], d['phases_ms_per_step'])"
for c in 3 4 5; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --pmc off --config $c > $OUT/bench_c$c.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_c$c.json')); print('BENCH c$c', round(d['value']), 'pts/s', d['roofline']['achieved'], d['phases_ms_per_step'])"; done

#!/bin/bash
# K3b locality experiment: does padding the leading dimension (breaking the 2^13-aligned row stride) raise the L2 hit rate?
TAG=${1:-ld}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
for mp in 5120 2048 8064; do timeout 300 ./tools/contract_ld_bench $mp 65536 > $OUT/ld_$mp.txt 2>&1; cat $OUT/ld_$mp.txt; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -o tcc -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/pmc_tcc.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- $REPO/tools/contract_ld_bench 5120 65536 > $OUT/pmc_fetch.txt 2>&1
cd $REPO
python - <<PY
import csv, glob
for name in ("tcc","fetch"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        rows=[r for r in csv.DictReader(open(f)) if "k_contract" in r["Kernel_Name"]]
        by={}
        for r in rows: by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"]); by[r["Dispatch_Id"]]["k"]=r["Kernel_Name"][:40]
        for d in sorted(by,key=int): print(name, d, by[d])
PY

#!/bin/bash
# Round 4: how much of k_contract's fabric traffic (2 x FETCH_SIZE: L2 misses, Infinity-Cache hits included) is HBM?  rocprofv3 on
# gfx950 lists no Infinity-Cache hit / miss counter (rocprofv3 --list-avail: TCC_EA0_RDREQ_DRAM counts every request "destined for DRAM
# (MC)", the cache sits behind that interface), so the split is made by LATENCY: TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ = average L2-miss
# latency of a kernel; tools/ea_probe gives the two pure cases under the same counters (6 GiB streamed once = HBM; 96 MiB re-read
# 40 x = Infinity Cache).  scripts/hbm_split.py turns the three averages into a fraction.  Usage (GPU box): bash scripts/gpu_hbm_split.sh [tag]
OUT=$PWD/gpurun_out/${1:-hbm_split}; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
CTR="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum"
timeout 200 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/probe -o probe -- $REPO/tools/ea_probe > $OUT/probe.txt 2> $OUT/probe.err
timeout 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/bench -o bench -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other > $OUT/bench.json 2> $OUT/bench.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu --pmc off --no-other > /dev/null 2>> $OUT/bench.err
cd $REPO
python scripts/hbm_split.py $OUT > $OUT/hbm_split.txt 2>&1
cat $OUT/probe.txt $OUT/hbm_split.txt
rm -rf $OUT/*/*.db 2>/dev/null

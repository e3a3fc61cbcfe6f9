timeout 300 ./tools/update_bench 8064 1 4 > gpurun_out/r05_update_bench.txt 2>&1; timeout 200 ./tools/update_bench 5120 1 >> gpurun_out/r05_update_bench.txt 2>&1
bash scripts/gpu_r05.sh tests
bash scripts/gpu_r05.sh bench

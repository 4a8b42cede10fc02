#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_update_div.py tests/test_gpu_split.py tests/test_gpu_multi.py -m gpu -q --timeout=600 -p no:cacheprovider \
   -k "cfg3_200_iterations_against or cfg4_full or restart_lanes or batched or restarts or multi" -s > gpurun_out/r03_t2.log 2>&1
rc=$?
grep -a "cfg3 \|cfg4\|batched restarts\|passed\|failed" gpurun_out/r03_t2.log | tail -20
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
NMF_RESTART_TRACE=1 timeout -k 10 200 python tools/restart_bench.py > gpurun_out/r03_restart_bench.log 2>&1; tail -70 gpurun_out/r03_restart_bench.log
timeout -k 10 200 python tools/single_split_sweep.py > gpurun_out/r03_single_split.log 2>&1; cat gpurun_out/r03_single_split.log
timeout -k 10 200 python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_cfg3_a.json 2> gpurun_out/r03_bench_cfg3_a.err; cat gpurun_out/r03_bench_cfg3_a.json | cut -c1-1500

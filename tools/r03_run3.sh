#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 60 python tools/malloc_probe.py > gpurun_out/r03_malloc_probe.log 2>&1; cat gpurun_out/r03_malloc_probe.log
NMF_RESTART_TRACE=1 timeout -k 10 200 python tools/restart_bench.py > gpurun_out/r03_restart_bench.log 2>&1; grep -a -v "lanes=1" gpurun_out/r03_restart_bench.log | tail -40
timeout -k 10 700 python -m pytest tests/test_gpu_update_div.py -m gpu -q --timeout=600 -p no:cacheprovider -k "cfg3_200_iterations_against or cfg4_full or restart_lanes" -s > gpurun_out/r03_t3.log 2>&1
rc=$?
grep -a "cfg3 \|cfg4\|passed\|failed\|Error" gpurun_out/r03_t3.log | tail -20
echo "pytest rc=$rc"

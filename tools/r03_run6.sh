#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_split.py tests/test_gpu_instantiations.py tests/test_gpu_multi.py tests/test_gpu_update_div.py -m gpu -q --timeout=600 -p no:cacheprovider \
   -k "batched or restart or instantiation or check or cfg5_full or multi" -s > gpurun_out/r03_t6.log 2>&1
rc=$?
grep -a "cfg5\|batched restarts\|passed\|failed\|Error" gpurun_out/r03_t6.log | tail -12
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
NMF_RESTART_TRACE=1 timeout -k 10 200 python tools/restart_bench.py > gpurun_out/r03_restart_bench.log 2>&1; grep -a "restarts x\|final KL" gpurun_out/r03_restart_bench.log | tail -32
timeout -k 10 200 python tools/boundary.py 1024x4096x64 4096x350x128 512x3445x30 2>&1 | cut -c1-260 > gpurun_out/r03_boundary.log; tail -12 gpurun_out/r03_boundary.log

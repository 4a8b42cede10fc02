#!/bin/bash
# GPU-box run 1 of round 3: the whole GPU suite, then the small-shape and batched-restart baselines
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=700 -p no:cacheprovider > gpurun_out/r03_t1.log 2>&1
rc=$?
tail -5 gpurun_out/r03_t1.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 150 python tools/small_bench.py > gpurun_out/r03_small_bench_base.log 2>&1 && tail -6 gpurun_out/r03_small_bench_base.log && \
timeout -k 10 300 python tools/restart_split_sweep.py > gpurun_out/r03_restart_sweep.log 2>&1; tail -40 gpurun_out/r03_restart_sweep.log

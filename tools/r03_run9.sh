#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_cli_tools.py tests/test_gpu_multi.py -m gpu -q --timeout=300 -p no:cacheprovider -k "cli_restarts or more_restarts" > gpurun_out/r03_t9.log 2>&1; tail -3 gpurun_out/r03_t9.log
bash tools/pmc_small.sh 2>&1 | tail -30

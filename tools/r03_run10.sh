#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_cli_tools.py tests/test_gpu_sharded.py -m gpu -q --timeout=300 -p no:cacheprovider > gpurun_out/r03_t10.log 2>&1; tail -5 gpurun_out/r03_t10.log

#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=700 -p no:cacheprovider > gpurun_out/r03_t8.log 2>&1
rc=$?
tail -4 gpurun_out/r03_t8.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/soak_small.py > gpurun_out/r03_soak_small.log 2>&1; cat gpurun_out/r03_soak_small.log | tail -8
timeout -k 10 200 python bench.py > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err; echo "bench rc=$? lines=$(wc -l < gpurun_out/r03_bench_final.json)"; cut -c1-330 gpurun_out/r03_bench_final.json

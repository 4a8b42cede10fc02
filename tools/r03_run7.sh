#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=700 -p no:cacheprovider -s > gpurun_out/r03_t7.log 2>&1
rc=$?
grep -a "cfg3 \|cfg4\|cfg5\|batched restarts\|passed\|failed\|Error" gpurun_out/r03_t7.log | tail -24
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/r03_profile.sh restarts 2>&1 | grep -v "nmf arena" | tail -40

#!/bin/bash
mkdir -p gpurun_out/r03_2rank
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout=700 -p no:cacheprovider > gpurun_out/r03_t5.log 2>&1
rc=$?
tail -6 gpurun_out/r03_t5.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
NMF_RESTART_TRACE=1 timeout -k 10 200 python tools/restart_bench.py > gpurun_out/r03_restart_bench.log 2>&1; grep -a "arena\|restarts x" gpurun_out/r03_restart_bench.log | grep -a -v "lanes=1" | tail -24
o=gpurun_out/r03_2rank
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --dist-backend gloo --same-device --comm auto --N 8192 --no-cpu-baseline > $o/auto.json 2> $o/auto.err
echo "rc=$? lines on stdout: $(wc -l < $o/auto.json)"; cut -c1-200 $o/auto.json; python3 -c "
import json; d=json.loads(open('$o/auto.json').read()); print({k:d.get(k) for k in ('allreduce_ms_per_step','compute_ms_per_step','comm','rccl','ms_per_step','n_gpus','scaling')})"
timeout -k 10 200 python bench.py --rehearse-sharded --steps 5 --warmup 2 --repeats 3 --N 8192 --no-cpu-baseline --comm rccl > $o/rccl_1rank.json 2> $o/rccl_1rank.err; echo "rc=$? lines on stdout: $(wc -l < $o/rccl_1rank.json)"

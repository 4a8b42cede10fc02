#!/bin/bash
mkdir -p gpurun_out/r03_2rank
export HSA_ENABLE_IPC_MODE_LEGACY=0
NMF_RESTART_TRACE=1 timeout -k 10 200 python tools/restart_bench.py > gpurun_out/r03_restart_bench.log 2>&1; grep -a "arena\|(4096,350,128)\|(1024,4096,64)" gpurun_out/r03_restart_bench.log | grep -a -v "lanes=1" | tail -24
o=gpurun_out/r03_2rank
timeout -k 10 200 python bench.py --rehearse-sharded --steps 5 --warmup 2 --repeats 3 --N 8192 --no-cpu-baseline --comm rccl > $o/rccl_1rank.json 2> $o/rccl_1rank.err; echo "rc=$?"; cut -c1-300 $o/rccl_1rank.json; python3 -c "
import json; d=json.loads(open('$o/rccl_1rank.json').read()); print({k:d.get(k) for k in ('allreduce_ms_per_step','compute_ms_per_step','allreduce_bytes','comm','rccl','ms_per_step')})"
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --dist-backend gloo --same-device --comm auto --N 8192 --no-cpu-baseline > $o/auto.json 2> $o/auto.err
echo "rc=$?"; python3 -c "
import json; d=json.loads(open('$o/auto.json').read()); print({k:d.get(k) for k in ('allreduce_ms_per_step','compute_ms_per_step','comm','rccl','ms_per_step','n_gpus','scaling')})"; grep -v "^$" $o/auto.err | tail -4
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --preset cfg4 --steps 3 --warmup 1 --repeats 2 --dist-backend gloo --same-device --comm torch --no-cpu-baseline > $o/cfg4_torch.json 2> $o/cfg4_torch.err
echo "rc=$?"; python3 -c "
import json; d=json.loads(open('$o/cfg4_torch.json').read()); print({k:d.get(k) for k in ('value','allreduce_ms_per_step','compute_ms_per_step','comm','ms_per_step','n_gpus','scaling')}, d['config']['workload'])"; grep -v "^$" $o/cfg4_torch.err | tail -3
timeout -k 10 700 python -m pytest tests/test_gpu_update_div.py -m gpu -q --timeout=600 -p no:cacheprovider -k "cfg3_200_iterations_against" -s > gpurun_out/r03_t4.log 2>&1
rc=$?
grep -a "cfg3 \|passed\|failed\|Error" gpurun_out/r03_t4.log | tail -12
echo "pytest rc=$rc"

cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/${ROUND:-r04}_2rank
mkdir -p $out
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --dist-backend gloo --same-device --comm torch --N 8192 --no-cpu-baseline > $out/torch.json 2> $out/torch.err
echo "rc=$?"; cut -c1-600 $out/torch.json; tail -3 $out/torch.err
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3 --dist-backend gloo --same-device --comm auto --N 8192 --no-cpu-baseline > $out/auto.json 2> $out/auto.err
echo "rc=$?"; cut -c1-600 $out/auto.json; grep -v "^$" $out/auto.err | tail -6

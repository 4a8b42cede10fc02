cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r02_prof1
mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -4 $out/tests.log
python3 bench.py > $out/bench.json 2> $out/bench.err; cut -c1-1500 $out/bench.json
# marker trace of an eager CLI run (roctx ranges per piece)
./nmf-gpu_amd/nmf generate --M 1024 --N 4096 --K 64 --X $out/X.bin --W $out/W.bin --H $out/H.bin > /dev/null
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $out/marker -- ./nmf-gpu_amd/nmf --X $out/X.bin --W $out/W.bin --H $out/H.bin --Wout $out/Wo.bin --Hout $out/Ho.bin --iters 50 --timers --thresh 1e-9 > $out/marker.log 2>&1
tail -3 $out/marker.log
rm -f $out/*.bin
find $out/marker -name "*stats*.csv" | head; find $out/marker -name "*marker*" | head

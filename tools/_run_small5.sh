cd /root/repo
timeout -k 10 300 python3 tools/split_check.py 256x384x64x10 1024x4096x64x20 4096x350x128x20 512x3445x30x20 100x77x5x10 333x1000x100x10 1000x260x64x10
timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 600 2>&1 | tail -5
export TMPDIR=/tmp
out=gpurun_out/r02_small3
mkdir -p $out
for cfg in "1024 4096 64 1 64" "4096 350 128 1 64" "512 3445 30 1 64"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $out/$tag -- python3 tools/small_iter.py $cfg > $out/$tag.log 2>&1
  python3 tools/trace_timeline.py $out/$tag 64 > $out/$tag.timeline 2>&1
  tail -1 $out/$tag.log; tail -5 $out/$tag.timeline
done
python3 tools/small_bench.py

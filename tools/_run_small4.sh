cd /root/repo
timeout -k 10 300 python3 tools/split_check.py 256x384x64x10 1024x4096x64x20 4096x350x128x20 512x3445x30x20 100x77x5x10 333x1000x100x10 512x512x64x10
echo "--- NW=4 forced"
NMF_SPLIT_NW=4 python3 tools/small_iter.py 1024 4096 64 1 400 1 2>&1 | grep "it/s"
NMF_SPLIT_NW=4 python3 tools/small_iter.py 512 3445 30 1 400 1 2>&1 | grep "it/s"
echo "--- NW auto"
for nw in 2 4 8; do python3 tools/small_iter.py 1024 4096 64 1 400 1 1 $nw; done 2>&1 | grep "it/s"
for nw in 4 7 14; do python3 tools/small_iter.py 512 3445 30 1 400 1 1 $nw; done 2>&1 | grep "it/s"
for nh in 8 11; do python3 tools/small_iter.py 4096 350 128 1 400 1 $nh 1; done 2>&1 | grep "it/s"
for b in 2 16; do python3 tools/small_iter.py 1024 4096 64 1 200 1 0 0 $b; done 2>&1 | grep "it/s"
export TMPDIR=/tmp
out=gpurun_out/r02_small2
mkdir -p $out
for cfg in "4096 350 128 0 64 1 8 1" "4096 350 128 0 64 1 11 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv -d $out/$tag -- python3 tools/small_iter.py $cfg > $out/$tag.log 2>&1
  python3 tools/trace_timeline.py $out/$tag 64 > $out/$tag.timeline 2>&1
  tail -1 $out/$tag.log; tail -5 $out/$tag.timeline
done

cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r02_small1
mkdir -p $out
for shp in "1024 4096 64" "4096 350 128" "512 3445 30"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/$tag -- python3 tools/small_iter.py $shp 0 64 1 > $out/$tag.log 2>&1
  python3 tools/trace_timeline.py $out/$tag 64 > $out/$tag.timeline 2>&1
  tail -1 $out/$tag.log; tail -6 $out/$tag.timeline
done
# split sweeps without the profiler
for nh in 1 2; do for nw in 2 4 8 16; do python3 tools/small_iter.py 1024 4096 64 1 400 1 $nh $nw; done; done 2>&1 | grep "it/s"
for nh in 4 8 11 16 32; do for nw in 1 2; do python3 tools/small_iter.py 4096 350 128 1 400 1 $nh $nw; done; done 2>&1 | grep "it/s"
for nh in 1 2; do for nw in 4 7 14 27; do python3 tools/small_iter.py 512 3445 30 1 400 1 $nh $nw; done; done 2>&1 | grep "it/s"
for b in 1 2 4 8 16; do python3 tools/small_iter.py 1024 4096 64 1 200 1 0 0 $b; done 2>&1 | grep "it/s"

cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r02_k64
mkdir -p $out
for shp in "4096 65536 64" "4096 65536 128"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/$tag -- python3 tools/small_iter.py $shp 1 24 > $out/$tag.log 2>&1
  python3 tools/trace_timeline.py $out/$tag 48 > $out/$tag.timeline 2>&1
  tail -1 $out/$tag.log; tail -6 $out/$tag.timeline
done

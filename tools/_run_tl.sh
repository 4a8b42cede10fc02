cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r02_prof4
mkdir -p $out
for shp in "1024 4096 64" "4096 350 128" "512 3445 30"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/small_$tag -- python3 tools/small_iter.py $shp 0 64 > $out/small_$tag.log 2>&1
  python3 tools/trace_timeline.py $out/small_$tag 64 > $out/small_$tag.timeline 2>&1
  tail -4 $out/small_$tag.timeline
done
find $out -name "*_agent_info.csv" -delete

#!/bin/bash
# kernel-trace of the batched loops (16 pairs per launch): which launch takes what
cd /root/repo; export TMPDIR=/tmp
out=gpurun_out/r03_batch_trace; mkdir -p $out
for shp in "4096 350 128" "1024 4096 64" "512 3445 30"; do
  tag=$(echo $shp | tr ' ' 'x')
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $out/$tag -- python3 tools/small_iter.py $shp 0 64 0 0 0 16 > $out/$tag.log 2>&1
  python3 tools/trace_timeline.py $out/$tag 96 > $out/$tag.timeline 2>&1; tail -1 $out/$tag.log; tail -5 $out/$tag.timeline
done
find $out -name "*_agent_info.csv" -delete

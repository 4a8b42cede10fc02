cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/profile_all
mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --timeout 600 > $out/tests.log 2>&1; tail -3 $out/tests.log
python3 bench.py > $out/bench.json 2> $out/bench.err; cut -c100-420 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
python3 tools/small_bench.py > $out/small_bench.log 2>&1; cat $out/small_bench.log
for shp in "1024 4096 64" "4096 350 128" "512 3445 30"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/small_$tag -- python3 tools/small_iter.py $shp 1 64 > $out/small_$tag.log 2>&1
  python3 tools/trace_timeline.py $out/small_$tag 64 > $out/small_$tag.timeline 2>&1
done
python3 tools/shape_bench.py 4096x65536x64 4096x65536x128 4096x65536x256 8192x16384x512 4096x65536x640 4096x65536x1024 4096x262144x256 > $out/shape_bench.log 2>&1; cat $out/shape_bench.log
python3 tools/restart_bench.py > $out/restart_bench.log 2>&1; cat $out/restart_bench.log
python3 tools/crossover.py > $out/crossover.log 2>&1; tail -25 $out/crossover.log
python3 tools/boundary.py 1024x4096x64 4096x350x128 512x3445x30 2>&1 | cut -c1-260 > $out/boundary.log; grep "200 iterations, use_graph=2\|use_graph=1" $out/boundary.log | head -4
python3 tools/check_bench.py
find $out -name "*_agent_info.csv" -delete

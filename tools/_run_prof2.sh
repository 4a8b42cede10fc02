cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/r02_prof2
mkdir -p $out
python3 bench.py > $out/bench.json 2> $out/bench.err; cut -c1-400 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "insts SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  set -- $grp; name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $out/pmc_$name.json 2> $out/pmc_$name.err
  echo "pmc $name done"
done
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_insts > $out/pmc_summary.txt 2>&1; head -30 $out/pmc_summary.txt
python3 tools/small_bench.py > $out/small_bench.log 2>&1; cat $out/small_bench.log
for shp in "1024 4096 64" "4096 350 128" "512 3445 30"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --output-format csv -d $out/small_$tag -- python3 tools/small_iter.py $shp 1 64 > $out/small_$tag.log 2>&1
  python3 tools/trace_timeline.py $out/small_$tag 64 > $out/small_$tag.timeline 2>&1
done
python3 tools/shape_bench.py 4096x65536x64 4096x65536x128 4096x65536x256 8192x16384x512 4096x65536x640 4096x65536x1024 > $out/shape_bench.log 2>&1; cat $out/shape_bench.log
python3 tools/restart_bench.py > $out/restart_bench.log 2>&1; cat $out/restart_bench.log
python3 bench.py --rehearse-sharded --comm rccl --no-cpu-baseline > $out/bench_rccl1.json 2> $out/bench_rccl1.err; cut -c100-330 $out/bench_rccl1.json
python3 bench.py --rehearse-sharded --comm torch --no-cpu-baseline > $out/bench_torch1.json 2> $out/bench_torch1.err; cut -c100-330 $out/bench_torch1.json
find $out -name "*_agent_info.csv" -delete
find $out/pmc_* -name "*.csv" -size +2M -delete

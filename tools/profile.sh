#!/bin/bash
# Regenerates what profiles/rNN_* holds: the bench line, its rocprofv3 kernel-trace stats, the PMC passes of the same command,
# small-shape bench lines + timelines, the batched-restart numbers, the true-K shape sweep.
#   gpurun -- 'ROUND=r05 bash tools/profile.sh [stage ...]'      (output under gpurun_out/$ROUND_profile; copy what is to be kept to profiles/)
cd /root/repo
export TMPDIR=/tmp
ROUND=${ROUND:-r05}
out=gpurun_out/${ROUND}_profile
mkdir -p $out
stages=${@:-bench kstats pmc small restarts shapes}   # also: pmck
for st in $stages; do case $st in
bench)
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 > $out/bench_cfg3.json 2> $out/bench_cfg3.err; cut -c1-400 $out/bench_cfg3.json ;;
kstats)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_prof.json 2> $out/bench_prof.err
  f=$(find $out/kstats -name "*kernel_stats.csv" | head -1); head -8 "$f" ;;
pmc)
  for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "insts SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
    set -- $grp; name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > $out/pmc_$name.log 2>&1
    echo "pmc $name done"
  done
  python3 tools/pmc_summary.py --json $out/pmc_static.json --shape 4096x65536x256 --source "profiles/${ROUND}_pmc_summary.md: rocprofv3 --pmc passes of \`python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline\` (FETCH_SIZE x 2 + WRITE_SIZE; not this run)" \
      $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_insts > $out/pmc_summary.txt 2>&1; cat $out/pmc_summary.txt ;;
pmck)
  # the same four counter passes on other K at 4096 x 65536 (PMCK="576 640 1024": the largest 64-column kernel, the wave-pair kernel)
  for K in ${PMCK:-576 640 1024}; do
    for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "insts SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
      set -- $grp; name=$1; shift
      timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmck${K}_$name -- python3 tools/small_iter.py 4096 65536 $K 0 4 -1 > $out/pmck${K}_$name.log 2>&1
    done
    { echo "# ${ROUND} PMC summary — 4096 x 65536 x $K (four --pmc passes of tools/small_iter.py 4096 65536 $K 0 4 -1; tools/pmc_summary.py)"; echo
      python3 tools/pmc_summary.py $out/pmck${K}_fetch $out/pmck${K}_write $out/pmck${K}_sq $out/pmck${K}_insts; } > $out/pmc_k$K.md 2>&1
    grep "fused_step_kernel.*SIMD-cycles" $out/pmc_k$K.md | cut -c1-260
  done ;;
small)
  for pr in cfg2 gold gold100 gold200 paper; do
    timeout -k 10 120 python3 bench.py --preset $pr --steps 200 --warmup 41 --cpu-budget 4 > $out/bench_$pr.json 2> $out/bench_$pr.err; cut -c100-330 $out/bench_$pr.json
  done
  timeout -k 10 120 python3 tools/small_bench.py > $out/small_bench.log 2>&1; cat $out/small_bench.log
  for shp in "1024 4096 64" "4096 350 128" "512 3445 30"; do
    tag=$(echo $shp | tr ' ' 'x')
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $out/small_$tag -- python3 tools/small_iter.py $shp 0 64 > $out/small_$tag.log 2>&1
    python3 tools/trace_timeline.py $out/small_$tag 64 > $out/small_$tag.timeline 2>&1; tail -5 $out/small_$tag.timeline
  done ;;
restarts)
  NMF_RESTART_TRACE=1 timeout -k 10 200 python3 tools/restart_bench.py > $out/restart_bench.log 2>&1; grep -v "nmf restarts" $out/restart_bench.log
  timeout -k 10 200 python3 tools/restart_split_sweep.py > $out/restart_sweep.log 2>&1; grep "whole call\|nsplit_h=0 nsplit_w=0" $out/restart_sweep.log ;;
shapes)
  timeout -k 10 600 python3 tools/shape_bench.py 4096x65536x48 4096x65536x64 4096x65536x96 4096x65536x100 4096x65536x128 4096x65536x160 4096x65536x192 4096x65536x200 4096x65536x224 4096x65536x256 \
      4096x65536x300 8192x16384x512 4096x65536x544 4096x65536x576 4096x65536x640 4096x65536x800 4096x65536x1024 4096x262144x256 4096x65536x16 4096x65536x32 > $out/shape_bench.log 2>&1; cat $out/shape_bench.log ;;
esac; done
find $out -name "*_agent_info.csv" -delete

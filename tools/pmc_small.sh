cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/${ROUND:-r04}_pmc_small
mkdir -p $out
for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "insts SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  set -- $grp; name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 tools/small_iter.py 1024 4096 64 0 24 > $out/pmc_$name.log 2>&1
  echo "pmc $name done"
done
python3 tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_insts > $out/pmc_summary.txt 2>&1; cat $out/pmc_summary.txt
find $out -name "*_agent_info.csv" -delete

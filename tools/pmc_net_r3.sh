#!/bin/bash
# MFMA utilisation, wave-cycle accounting, LDS conflicts and the effective clock of the net forward's kernels (11x11, 4096 positions):
# SQ / GRBM counters in SEPARATE rocprofv3 --pmc passes (kernel trace only) over tools/probe_net_min.py (value branch on the main
# stream so that kernels do not overlap), mean over the dispatches of each kernel -> gpurun_out/pmc_net_r3.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_net_r3.txt; rm -f $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1)); d=/tmp/pnet3_$i; rm -rf $d
  S=${S:-11} B=4096 N=12 BRANCH=0 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python /root/repo/tools/probe_net_min.py > $d.log 2>&1 || echo "## set $i rc=$?" >> $OUT
  DB=$(find $d -name "*.db" 2>/dev/null | head -1)
  if [ -n "$DB" ]; then echo "## pass: $set" >> $OUT; python /root/repo/tools/pmc_summary.py $DB af_ 8 >> $OUT 2>/dev/null; fi
  # kernel durations of the same pass (ns), for the effective clock = GRBM_GUI_ACTIVE / duration
  [ -n "$DB" ] && [ $i -eq 3 ] && python /root/repo/tools/rocpd_stats.py $DB 16 >> $OUT
  rm -rf $d
done
cat $OUT

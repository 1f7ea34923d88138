#!/bin/bash
# usage: tools/pmc_modes.sh "<mode list>" -- one rocprofv3 --pmc pass per counter set and conv mode (lean probe)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc2; mkdir -p $OUT
for mode in $1; do
 i=0
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" "TA_BUSY_avr SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
  i=$((i+1)); d=/tmp/pm_${mode}_$i
  MODE=$mode B=4096 N=2 timeout 120 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python /root/repo/tools/probe_net_min.py > $d.log 2>&1 || echo "set $i mode $mode rc=$?"
  DB=$(find $d -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB | grep -A5 "af_conv_wino" >> $OUT/mode$mode.txt
 done
 echo "== mode $mode"; cat $OUT/mode$mode.txt
done

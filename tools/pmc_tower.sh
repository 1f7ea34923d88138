#!/bin/bash
# one rocprofv3 --pmc pass per counter set over the lean tower probe; prints the af_tower_conv rows
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tower.txt; rm -f $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); d=/tmp/pt_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python /root/repo/tools/probe_tower_min.py > $d.log 2>&1 || echo "set $i rc=$?"
  DB=$(find $d -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB 2>/dev/null | grep -v "^#" | grep -A5 "af_tower_conv" >> $OUT
done
cat $OUT

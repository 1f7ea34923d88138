#!/bin/bash
# where the tick kernel's time goes: SQ instruction / wait counters (separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tick.txt; rm -f $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=/tmp/ptk_$i
  TICKS=40 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python /root/repo/tools/probe_tick_min.py > $d.log 2>&1 || echo "pass $i rc=$?"
  DB=$(find $d -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB af_tick 2>/dev/null >> $OUT
done
cat $OUT

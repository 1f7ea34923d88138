#!/bin/bash
# Where the steady-state tick kernel's time goes (config 2, 4096 games, after ~3.5 plies per game so that terminals, the
# collector and restarts run): SQ instruction / wait counters in SEPARATE rocprofv3 --pmc passes (kernel trace only), means
# over the last 400 dispatches -> gpurun_out/pmc_tick_r2.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tick_r2.txt; rm -f $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32"; do
  i=$((i+1)); d=/tmp/ptk2_$i; rm -rf $d
  for ticks in ${TICKS:-1500} 600; do
    TICKS=$ticks timeout 600 rocprofv3 --kernel-trace --pmc $set -d $d -o p -- python /root/repo/tools/probe_tick_min.py > $d.log 2>&1
    DB=$(find $d -name "*.db" 2>/dev/null | head -1)
    if [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB af_tick 400 > $d.txt 2>/dev/null && [ -s $d.txt ]; then
      echo "## pass: $set (ticks $ticks)" >> $OUT; cat $d.txt >> $OUT; break
    fi
    echo "## pass '$set' with $ticks ticks failed, retrying shorter" >> $OUT
  done
  rm -rf $d
done
cat $OUT

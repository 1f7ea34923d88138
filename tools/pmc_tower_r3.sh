#!/bin/bash
# Is the bf16 tower power-bound?  MFMA-busy cycles and the effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) of af_tower_conv for
# the full kernel and for its run-time ablations (ABL bits: 1 no staging, 2 no stores / epilogue; ABLS="0 3 0"), one
# rocprofv3 --pmc pass (kernel trace only) per ablation over tools/probe_tower_min.py -> gpurun_out/pmc_tower_r3.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tower_r3.txt; rm -f $OUT
for abl in ${ABLS:-0 3 0}; do
  d=/tmp/ptw3_$abl; rm -rf $d
  ABL=$abl BLOCKS=8 N=6 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $d -o p -- python /root/repo/tools/probe_tower_min.py > $d.log 2>&1 || echo "## abl $abl rc=$?" >> $OUT
  DB=$(find $d -name "*.db" 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    echo "## ablation bits: $abl" >> $OUT
    python /root/repo/tools/pmc_summary.py $DB af_tower_conv 64 >> $OUT 2>/dev/null
    python /root/repo/tools/rocpd_stats.py $DB 4 | grep -i "af_tower_conv\|Name" >> $OUT
  fi
  rm -rf $d
done
cat $OUT

#!/bin/bash
# bf16 tower (config 5): is LDS what limits af_tower_conv?  LDS instruction / bank-conflict / busy counters next to the MFMA-busy share
# (VERDICT r4 next #2), separate rocprofv3 --pmc passes (kernel trace only) over tools/probe_tower_min.py -> gpurun_out/pmc_tower_r5.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tower_r5.txt; rm -f $OUT
AVAIL=/tmp/avail_r5.txt
rocprofv3 -L > $AVAIL 2>&1 || rocprofv3 --list-avail > $AVAIL 2>&1
echo "# LDS-related counters this rocprofv3 offers:" >> $OUT
grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INST_CYCLES_[A-Z_]*\|SQ_WAIT_INST_ANY\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_[A-Z_]*" $AVAIL | sort -u | tr '\n' ' ' >> $OUT; echo >> $OUT
have() { grep -q "\b$1\b" $AVAIL; }
pass() {  # name, counters...
  local tag=$1; shift
  local ctrs=""
  for c in "$@"; do if have $c; then ctrs="$ctrs $c"; else echo "## ($c not offered)" >> $OUT; fi; done
  [ -z "$ctrs" ] && return
  local d=/tmp/ptw5_$tag; rm -rf $d
  ENGINE=${ENGINE:-0} BLOCKS=8 N=6 timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $d -o p -- python /root/repo/tools/probe_tower_min.py > $d.log 2>&1 || { echo "## pass $tag ($ctrs) rc=$?" >> $OUT; tail -3 $d.log >> $OUT; }
  local DB=$(find $d -name "*.db" 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    echo "## pass $tag:$ctrs" >> $OUT
    python /root/repo/tools/pmc_summary.py $DB af_tower_conv 64 >> $OUT 2>/dev/null
    python /root/repo/tools/rocpd_stats.py $DB 4 | grep -i "af_tower_conv\|Name" >> $OUT
  fi
  rm -rf $d
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass lds1 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass lds2 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_LDS GRBM_GUI_ACTIVE
pass lds3 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ATOMIC_RETURN SQ_LDS_MEM_VIOLATIONS GRBM_GUI_ACTIVE
pass inst SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
cat $OUT

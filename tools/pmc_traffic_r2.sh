#!/bin/bash
# HBM traffic per dispatch of the config-2 tick at STEADY STATE (tick kernel + the fp16 split-operand net): FETCH_SIZE, WRITE_SIZE and
# the L2 hit counters in SEPARATE rocprofv3 --pmc passes (kernel trace only); 3500 ticks (~8 plies per game: terminals, collector,
# restarts are running), means over the last 400 dispatches of each kernel -> gpurun_out/pmc_traffic_r2.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_traffic_r2.txt; rm -f $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/ptr2_$(echo $c | tr ' ' '_')
  TICKS=${TICKS:-3500} timeout 900 rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python /root/repo/tools/probe_tick_min.py > $d.log 2>&1 || echo "pass $c rc=$?"
  DB=$(find $d -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB af_ 400 2>/dev/null >> $OUT
  tail -1 $d.log >> $OUT
  rm -rf $d
done
cat $OUT

#!/bin/bash
# HBM traffic per dispatch of the config-2 tick (tick kernel + the fp16 split-operand net): FETCH_SIZE, WRITE_SIZE and the L2 hit
# counters in SEPARATE rocprofv3 --pmc passes (kernel trace only), means over the last 400 dispatches of each kernel
# -> gpurun_out/pmc_traffic_r4.txt.  TICKS (default 3500 ~ 8 plies per game: terminals, collector and restarts are running) can be
# lowered per pass: rocprofv3 7.2 sometimes crashes at exit on long counter runs, so every pass is retried with fewer ticks.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_traffic_r4.txt; rm -f $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  for ticks in ${TICKS:-3500} 1500 600; do
    d=/tmp/ptr4_$(echo $c | tr ' ' '_'); rm -rf $d
    TICKS=$ticks timeout 900 rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python /root/repo/tools/probe_tick_min.py > $d.log 2>&1
    DB=$(find $d -name "*.db" 2>/dev/null | head -1)
    if [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB af_ 400 > $d.txt 2>/dev/null && [ -s $d.txt ]; then
      echo "## pass: $c, ticks: $ticks" >> $OUT; cat $d.txt >> $OUT; grep counters $d.log | tail -1 >> $OUT; break
    fi
    echo "## pass $c with $ticks ticks failed (rocprofv3 rc), retrying shorter" >> $OUT
  done
  rm -rf $d
done
cat $OUT

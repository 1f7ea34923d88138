#!/bin/bash
# HBM traffic per dispatch of the config-2 tick (tick kernel + fp32 net): FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (kernel trace only), means per kernel -> gpurun_out/pmc_traffic.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_traffic.txt; rm -f $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/ptr_$(echo $c | tr ' ' '_')
  TICKS=60 timeout 200 rocprofv3 --kernel-trace --pmc $c -d $d -o p -- python /root/repo/tools/probe_tick_min.py > $d.log 2>&1 || echo "pass $c rc=$?"
  DB=$(find $d -name "*.db" | head -1)
  [ -n "$DB" ] && python /root/repo/tools/pmc_summary.py $DB af_ 2>/dev/null >> $OUT
done
cat $OUT

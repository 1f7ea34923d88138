#!/bin/bash
# per-kernel durations of the 15x15 (S=15) or 11x11 forward for libaf_net variants: rocprofv3 --kernel-trace over tools/probe_net_min.py
# usage: S=15 tools/prof_f16s_layers.sh r4 default   -> gpurun_out/prof_f16s_layers_S<S>.txt
cd /tmp && export TMPDIR=/tmp
S=${S:-15}
OUT=/root/repo/gpurun_out/prof_f16s_layers_S$S.txt; rm -f $OUT
for v in "$@"; do
  if [ $v = default ]; then unset AF_NET_LIB; else export AF_NET_LIB=/root/repo/alphafive_amd/_lib/variants/libaf_net_$v.so; fi
  d=/tmp/pfl_$v; rm -rf $d
  S=$S N=${N:-12} ABLBITS=${ABLBITS:-0} timeout 300 rocprofv3 --kernel-trace -d $d -o p -- python /root/repo/tools/probe_net_min.py > $d.log 2>&1 || echo "## $v rc=$?" >> $OUT
  DB=$(find $d -name "*.db" | head -1)
  echo "## $v (S=$S, ABLBITS=${ABLBITS:-0})" >> $OUT
  [ -n "$DB" ] && python /root/repo/tools/rocpd_stats.py $DB 24 | cut -c1-200 >> $OUT
  rm -rf $d
done
cat $OUT

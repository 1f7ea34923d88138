#!/bin/bash
# A/B of libaf_tower.so builds in their own processes (AF_TOWER_LIB): "new" = the in-tree build, others = alphafive_amd/_lib/variants/libaf_tower_<name>.so
# usage: VARIANTS="old new" TUNES="2:0;0:8;0:16" bash tools/tower_ab.sh
for r in 1 2; do
for v in ${VARIANTS:-new}; do
  if [ $v = new ]; then unset AF_TOWER_LIB; else export AF_TOWER_LIB=/root/repo/alphafive_amd/_lib/variants/libaf_tower_$v.so; fi
  echo "== $v (round $r)"; TUNES="${TUNES:-2:0;2:3}" timeout 300 python tools/probe_tower.py 2>&1 | grep -E "^hip vs|^tower hip|^net hip:|^tune"
done; done

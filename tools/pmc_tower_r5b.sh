#!/bin/bash
# bf16 tower: MFMA-busy share, effective clock and wait shares of engine 0 (af_tower_conv) / 2 (af_tower_conv3 for both convolutions, r5 scalar
# epilogue stages) / optionally the r4 build of engine 2 (AF_TOWER_LIB) -> gpurun_out/pmc_tower_r5b.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tower_r5b.txt; rm -f $OUT
run() {  # tag engine lib counters...
  local tag=$1 eng=$2 lib=$3; shift 3
  local d=/tmp/ptw5b_$tag; rm -rf $d
  if [ -n "$lib" ]; then export AF_TOWER_LIB=$lib; else unset AF_TOWER_LIB; fi
  ENGINE=$eng BLOCKS=8 N=6 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $d -o p -- python /root/repo/tools/probe_tower_min.py > $d.log 2>&1 || echo "## $tag rc=$?" >> $OUT
  local DB=$(find $d -name "*.db" 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    echo "## $tag engine $eng lib ${lib:-in-tree}: $*" >> $OUT
    python /root/repo/tools/pmc_summary.py $DB af_tower_conv 64 >> $OUT 2>/dev/null
    python /root/repo/tools/rocpd_stats.py $DB 4 | grep -i "af_tower_conv\|Name" >> $OUT
  fi
  rm -rf $d
}
R4=/root/repo/alphafive_amd/_lib/variants/libaf_tower_r4.so
for rep in 1 2; do
run e0_mfma 0 "" SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run e2_mfma 2 "" SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
[ -f $R4 ] && run e2r4_mfma 2 $R4 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES
done
run e2_wait 2 "" SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
run e2_lds 2 "" SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
python3 - <<'PY' >> $OUT
import re
t = open("/root/repo/gpurun_out/pmc_tower_r5b.txt").read()
print("# summary: pass | kernel | us | GHz = GRBM_GUI_ACTIVE / 8 / duration | MFMA busy = busy cycles / (1024 SIMDs x active cycles) | busy x GHz")
for sec in t.split("## ")[1:]:
    head = sec.splitlines()[0]
    if "mfma" not in head.split()[0]:
        continue
    cur, ctr = None, {}
    for line in sec.splitlines()[1:]:
        if line.startswith("void"):
            name = re.sub(r"\(TowerArgs\).*", "", line.replace("void ", "")).strip()
            f = line.split()
            if len(f) > 6 and f[-1].endswith("%"):
                ctr.setdefault(name, {})["avg_ns"] = float(f[-4])
            else:
                cur = name
                ctr.setdefault(cur, {})
        m = re.match(r"\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
        if m and cur:
            ctr[cur][m.group(1)] = float(m.group(2))
    for k, c in ctr.items():
        if "avg_ns" in c and "GRBM_GUI_ACTIVE" in c:
            act = c["GRBM_GUI_ACTIVE"] / 8
            ghz = act / c["avg_ns"]
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * act)
            print("%-28s %-32s %7.1f us  %.3f GHz  busy %.3f  busy x GHz %.3f" % (head.split(":")[0], k, c["avg_ns"] / 1e3, ghz, busy, busy * ghz))
PY
cat $OUT

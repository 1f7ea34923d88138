#!/bin/bash
# bf16 tower: MFMA-busy share and effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) of the convolution kernels of engine 0
# (af_tower_conv) and engine 2 (af_tower_conv3: epilogue under the other tile pair's MFMAs), one rocprofv3 --pmc pass (kernel trace
# only) per engine over tools/probe_tower_min.py -> gpurun_out/pmc_tower_r4.txt
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_tower_r4.txt; rm -f $OUT
for eng in ${ENGINES:-0 2 0 2}; do
  d=/tmp/ptw4_$eng; rm -rf $d
  ENGINE=$eng BLOCKS=8 N=6 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU -d $d -o p -- python /root/repo/tools/probe_tower_min.py > $d.log 2>&1 || echo "## engine $eng rc=$?" >> $OUT
  DB=$(find $d -name "*.db" 2>/dev/null | head -1)
  if [ -n "$DB" ]; then
    echo "## engine $eng" >> $OUT
    python /root/repo/tools/pmc_summary.py $DB af_tower_conv 64 >> $OUT 2>/dev/null
    python /root/repo/tools/rocpd_stats.py $DB 4 | grep -i "af_tower_conv\|Name" >> $OUT
  fi
  rm -rf $d
done
python3 - <<'PY' >> $OUT
import re
t = open("/root/repo/gpurun_out/pmc_tower_r4.txt").read()
print("# summary: kernel | us | GHz = GRBM_GUI_ACTIVE / 8 / duration | MFMA busy = busy cycles / (1024 SIMDs x active cycles) | busy x GHz")
for sec in t.split("## engine ")[1:]:
    eng = sec.split()[0]
    ctr = {}
    cur = None
    for line in sec.splitlines():
        if line.startswith("void"):
            cur = re.sub(r"\(TowerArgs\).*", "", line.replace("void ", "")).strip()
            if "Calls" not in line and not re.search(r"\d+\s+\d+\s+\d+\s+\d+\s+\d+", line):
                ctr.setdefault(cur, {})
                continue
            f = line.split()
            name = cur
            ctr.setdefault(name, {})["avg_ns"] = float(f[-4])
        m = re.match(r"\s+(\S+)\s+n=\d+\s+mean=(\S+)", line)
        if m and cur:
            ctr[cur][m.group(1)] = float(m.group(2))
    for k, c in ctr.items():
        if "avg_ns" in c and "GRBM_GUI_ACTIVE" in c:
            act = c["GRBM_GUI_ACTIVE"] / 8
            ghz = act / c["avg_ns"]
            busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * act)
            print("engine %s %-34s %7.1f us  %.3f GHz  busy %.3f  busy x GHz %.3f  VALU insts %.3g" % (eng, k, c["avg_ns"] / 1e3, ghz, busy, busy * ghz, c.get("SQ_INSTS_VALU", 0)))
PY
cat $OUT

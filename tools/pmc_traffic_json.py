#!/usr/bin/env python
"""gpurun_out/pmc_traffic_r4.txt (tools/pmc_traffic_r4.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC hit+miss in separate
passes) -> profiles/r4_pmc_hbm_traffic.json, the file bench.py's roofline.traffic is read from.
usage: python tools/pmc_traffic_json.py [in.txt] [out.json]"""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "pmc_traffic_r4.txt")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "profiles", "r4_pmc_hbm_traffic.json")


def short(name):
    """kernel name without namespaces and argument list, template arguments kept (balanced brackets)"""
    name = re.sub(r"\(anonymous namespace\)::", "", name.strip())
    name = re.sub(r"^void ", "", name)
    m = re.match(r"[A-Za-z_0-9]+", name)
    if not m:
        return name
    out, i = m.group(0), m.end()
    if i < len(name) and name[i] == "<":
        depth = 0
        for j in range(i, len(name)):
            depth += name[j] == "<"
            depth -= name[j] == ">"
            if depth == 0:
                return out + name[i:j + 1].replace(" >", ">")
        return out + name[i:]                     # (truncated by the producer of the text file)
    return out


kern, passes, cur = {}, [], None
for line in open(src):
    if line.startswith("## pass:"):
        passes.append(line[3:].strip())
        continue
    if line.startswith("##") or line.startswith("#") or not line.strip():
        continue
    if not line.startswith(" "):
        cur = None if line.startswith("counters") else kern.setdefault(short(line), {})      # (the probe's own last stdout line)
        continue
    m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)(\s+dispatches=(\d+))?", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(3))
        if m.group(5):
            cur.setdefault("_dispatches", {})[m.group(1)] = int(m.group(5))

tick = next(k for k in kern if k.startswith("af_tick_kernel"))
out = {}
net_raw = net_corr = 0.0
for k, c in kern.items():
    hit, miss = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
    ent = {"FETCH_SIZE": c.get("FETCH_SIZE"), "WRITE_SIZE": c.get("WRITE_SIZE"),
           "l2_hit_rate": round(hit / (hit + miss), 3) if hit is not None and hit + miss > 0 else None}
    d = c.get("_dispatches", {})
    per_tick = 1
    if "FETCH_SIZE" in d and "FETCH_SIZE" in kern[tick].get("_dispatches", {}):
        per_tick = round(d["FETCH_SIZE"] / kern[tick]["_dispatches"]["FETCH_SIZE"])
    ent["launches_per_forward"] = per_tick
    out[k] = ent
    if k != tick and ent["FETCH_SIZE"] is not None and ent["WRITE_SIZE"] is not None and per_tick >= 1 and not k.startswith("af_pack"):
        net_raw += per_tick * (ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
        net_corr += per_tick * (2 * ent["FETCH_SIZE"] + ent["WRITE_SIZE"]) * 1024
t = out[tick]
doc = {
    "source": "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc TCC_HIT_sum TCC_MISS_sum (three separate passes, --kernel-trace only; "
              "tools/pmc_traffic_r4.sh) over tools/probe_tick_min.py: config 2 (4096 games, 11x11); counter unit KB per dispatch, mean over "
              "the last 400 dispatches of each kernel; assembled by tools/pmc_traffic_json.py. Passes: " + "; ".join(passes),
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read "
                  "(16 B/lane, global_load and LDS-DMA alike) -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; WRITE_SIZE and mixed-width "
                  "kernels are uncalibrated, so the raw sum (FETCH_SIZE + WRITE_SIZE) * 1024 is listed too",
    "workload": {"games": 4096, "board_size": 11},
    "kernels_kb_per_dispatch": out,
    "net_forward_bytes_per_launch": {"raw": int(net_raw), "corrected": int(net_corr),
                                     "kernels": [k for k in out if k != tick and not k.startswith("af_pack")]},
    "net_forward_algorithmic_bytes_per_launch": None,
    "tick_kernel_bytes_per_launch": {"raw": int((t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024),
                                     "corrected": int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024), "kernel": tick},
}
sys.path.insert(0, REPO)
from alphafive_amd import net_hip            # noqa: E402
doc["net_forward_algorithmic_bytes_per_launch"] = 4096 * net_hip.roofline_info(11)["algorithmic_bytes_per_position"]
doc["reading"] = ("net forward: %.2f GB from the counters against %.2f GB algorithmic (one read of every input slab incl. the projection "
                  "inputs, one write of every output; alphafive_amd/net_hip.py): no wasted re-reads" % (net_corr / 1e9, doc["net_forward_algorithmic_bytes_per_launch"] / 1e9))
json.dump(doc, open(dst, "w"), indent=1)
print("net forward: %.3f GB corrected (%.3f raw); tick kernel: %.1f MB corrected" % (net_corr / 1e9, net_raw / 1e9, doc["tick_kernel_bytes_per_launch"]["corrected"] / 1e6))

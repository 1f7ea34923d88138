"""Sample the GPU's power sensor and shader clock (amdgpu hwmon / sysfs, ~20 Hz) while a command runs.
usage: python tools/power_trace.py OUT.csv -- <command ...>
Prints one JSON line: the power cap, and mean / p50 / p95 / max socket power and mean shader clock over the busy part of the run
(samples with gpu_busy_percent >= 90).  Evidence for DESIGN's "the forward sits at the power wall" (clock figures there come
from counters; this is the sensor itself)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _first(pats):
    for p in pats:
        hits = sorted(glob.glob(p))
        if hits:
            return hits[0]
    return None


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def visible_gpu_bdf():
    """PCI address of HIP device 0 as THIS environment sees it (a box may carry more GPUs than the one it hands us)."""
    code = ("import torch; p = torch.cuda.get_device_properties(0); "
            "print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))")
    try:
        return subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode().split()[-1]
    except Exception:
        return None


def sensors():
    bdf = visible_gpu_bdf()
    hw = _first(["/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf]) if bdf else None
    if hw is None:
        return None
    dev = "/sys/bus/pci/devices/%s" % bdf
    return {"bdf": bdf, "power": _first([hw + "/power1_average", hw + "/power1_input"]), "cap": _first([hw + "/power1_cap"]),
            "cap_max": _first([hw + "/power1_cap_max"]), "sclk": _first([hw + "/freq1_input"]),
            "busy": _first([dev + "/gpu_busy_percent"]), "temp": _first([hw + "/temp1_input"])}


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    s = sensors()
    if s is None or s["power"] is None:
        print(json.dumps({"error": "no amdgpu hwmon power sensor under /sys/class/drm", "sensors": s}))
        return subprocess.call(cmd)
    rows, stop = [], threading.Event()

    def sample():
        while not stop.is_set():
            p, f, b = _read(s["power"]), _read(s["sclk"]) if s["sclk"] else None, _read(s["busy"]) if s["busy"] else None
            rows.append((time.time(), float(p) / 1e6 if p else float("nan"), float(f) / 1e6 if f else float("nan"),
                         float(b) if b else float("nan")))
            time.sleep(0.05)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t0 = time.time()
    rc = subprocess.call(cmd)
    stop.set()
    th.join()
    with open(out, "w") as f:
        f.write("t_s,power_w,sclk_mhz,busy_pct\n")
        for t, p, c, b in rows:
            f.write("%.3f,%.1f,%.0f,%.0f\n" % (t - t0, p, c, b))
    pw = [r[1] for r in rows if r[1] == r[1]]
    top = max(pw) if pw else 0.0
    have_busy = any(r[3] == r[3] for r in rows)
    sel = [r for r in rows if r[1] == r[1] and ((r[3] >= 90.0) if have_busy else (r[1] > 0.8 * top))]
    busy = sorted(r[1] for r in sel)
    clk = [r[2] for r in sel if r[2] == r[2]]
    idle = sorted(pw)[:max(1, len(pw) // 20)]

    def pct(v, q):
        return v[min(len(v) - 1, int(q * len(v)))] if v else None
    cap, capmax = _read(s["cap"]) if s["cap"] else None, _read(s["cap_max"]) if s["cap_max"] else None
    print(json.dumps({"cmd": " ".join(cmd)[:160], "rc": rc, "samples": len(rows), "busy_samples": len(busy),
                      "power_cap_w": float(cap) / 1e6 if cap else None, "power_cap_max_w": float(capmax) / 1e6 if capmax else None,
                      "busy_power_w": {"mean": sum(busy) / len(busy) if busy else None, "p50": pct(busy, 0.5), "p95": pct(busy, 0.95),
                                       "max": top}, "idle_power_w": sum(idle) / len(idle) if idle else None,
                      "busy_sclk_mhz_mean": sum(clk) / len(clk) if clk else None, "sensor": s["power"], "bdf": s["bdf"]}))
    return rc


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""bench.py — self-play moves/sec (BASELINE.json metric) on N MI355X of one node.

    python bench.py                              # = --gpus 1 --steps 20 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: 4096 concurrent 11x11 games per GPU, 500 sims/move
(cap 642), training-mode search (Dirichlet noise, forced root visits, temperature play),
alphaFive-6960 weights, fp32 net, batched leaf evaluation.  Weak scaling: every rank owns
its own 4096 games (game id = rank*4096 + g), no collective on the tick path; finished
episodes are gathered to rank 0 over RCCL at step boundaries.

A "step" = one pass of the hot path over the batch that commits one move per game on
average: the rank ticks (tree kernel -> leaf batch -> net) until its games have committed
G more plies (~400-500 ticks).  value = plies committed by all ranks in the timed region /
max-over-ranks wall time.  The metric is a STEADY-STATE rate: the defaults warm up for
8 plies per game and time 20 more, so finished episodes, store collection and game restarts are
inside the timed region; a run in which no episode finished is reported with "value": null.  Inputs are synthetic (all games start from the empty board) and
resident in HBM; nothing crosses PCIe on the tick path.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_POSITION = 118_727_264          # SURVEY §2.2 (59,363,632 MAC) at S=11
PEAK_FP32_MFMA_TFLOPS = 157.3            # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E spec peak


def make_cfg(sims=500, upper=642, board=11):
    return types.SimpleNamespace(board_size=board, goal=5, simulation_per_step=sims, upper_simulation_per_step=upper,
                                 init_temp=1.2, gamma=0.94, tau_decay_rate=0.94, tau_decay_rate_r=0.9,
                                 dirichlet_alpha=0.3, c_puct=5.0)


def tree_bytes(ct, C):
    """SURVEY §8d algorithmic bytes: per select 12L+40 (+16 backup), per expand 4L+32+16C+4, per sim R."""
    R = 128 if C <= 128 else 256
    return (12 * ct["legal_sum"] + 56 * ct["selects"] + 4 * ct["legal_sum_expand"]
            + (32 + 16 * C + 4) * ct["expands"] + R * ct["sims"])


def cpu_baseline(cfg, weights, budget_s=15.0, game_id=0, training=True, whole_game=False):
    """The C oracle (oracle/af_oracle.c, "port") + torch-CPU fp32 net on ONE host core.  training=True: the config-2
    search for one game (plies until the time budget is spent, or the complete episode with whole_game).
    training=False: BASELINE config 1 — the eval-mode loop of self_play.py:79-106 (last_action=None at every root, tree
    kept and shared by both colours) for the whole game."""
    import torch
    import oracle
    from alphafive_amd.network import ResNet
    torch.set_num_threads(1)
    net = ResNet(cfg.board_size, device="cpu", seed=0)
    if weights:
        net.load_npz(weights)
    pl = oracle.OraclePlayer(cfg, training=training, rng_mode=oracle.RNG_PHILOX, seed=0, game_id=game_id, pv_fn=net.eval)
    board = np.zeros((cfg.board_size, cfg.board_size), np.int8)
    state, last, plies, over = oracle.board_to_state(board), None, 0, False
    t0 = time.time()
    while whole_game or time.time() - t0 < budget_s:
        _, act, _ = pl.get_action(state, last if training else None)
        board = oracle.step(oracle.state_to_board(state, cfg.board_size), act)
        state, last, plies = oracle.board_to_state(board), act, plies + 1
        if oracle.is_game_over(board, cfg.goal)[0]:
            over = True
            break
        if whole_game and time.time() - t0 > 4 * budget_s:      # safety net for a pathological game
            break
    dt = time.time() - t0
    st = pl.stats()
    pl.close()
    what = ("complete %d-ply game" % plies) if over else ("first %d plies" % plies)
    mode = "training mode (config 2 search)" if training else "eval mode (config 1: self_play.py loop)"
    return {"value": plies / dt, "unit": "moves/s", "cores": 1, "kind": "port", "plies": plies, "dt": dt, "complete": over,
            "sample": f"1 game, {what}, 11x11, {cfg.simulation_per_step} sims/move, {mode}, C oracle + torch-CPU fp32 net, "
                      f"1 thread, {dt:.1f} s, {st['sims']} sims / {st['expands']} net evals"}


def cpu_tree_only(cfg, budget_s=3.0):
    """Tree-only search rate of the C oracle with its built-in integer pseudo-net (no network time): sims/s, directly
    comparable with the 341 sims/s of the reference's Python tree."""
    import oracle
    pl = oracle.OraclePlayer(cfg, training=True, rng_mode=oracle.RNG_PHILOX, seed=0, game_id=0, pseudo_salt=777, pseudo_peak=8192)
    board = np.zeros((cfg.board_size, cfg.board_size), np.int8)
    state, last = oracle.board_to_state(board), None
    t0 = time.time()
    while time.time() - t0 < budget_s:
        _, act, _ = pl.get_action(state, last)
        board = oracle.step(oracle.state_to_board(state, cfg.board_size), act)
        state, last = oracle.board_to_state(board), act
        if oracle.is_game_over(board, cfg.goal)[0]:
            break
    dt = time.time() - t0
    sims = pl.stats()["sims"]
    pl.close()
    return sims / dt


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                tok = f.read().split()
            if path.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(float(tok[0]) / float(tok[1]))))
            else:
                q = int(tok[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_all_cores(cfg, weights, budget_s=9.0, max_workers=64):
    """one game per host core, every worker the same 1-thread C oracle + torch-CPU net as
    cpu_baseline(), each playing one COMPLETE training-mode episode (cut at 4 x budget_s); aggregate moves/s = all
    plies / the slowest worker's time."""
    import subprocess
    n = max(1, min(usable_cores(), max_workers))
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", AF_CPU_WORKER="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget_s), "--sims", str(cfg.simulation_per_step),
           "--upper", str(cfg.upper_simulation_per_step), "--board", str(cfg.board_size)]
    procs = [subprocess.Popen(cmd + ["--seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for i in range(n)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s * 6 + 120)
            res.append(json.loads(out.decode().strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if not res:
        return None
    plies, dt = sum(r["plies"] for r in res), max(r["dt"] for r in res)
    done = sum(1 for r in res if r.get("complete"))
    return {"value": plies / dt, "unit": "moves/s", "cores": len(res), "host_cores": os.cpu_count(), "usable_cores": usable_cores(),
            "sample": f"{len(res)} processes x 1 game x 1 thread, one complete training-mode episode each ({done} finished, the rest "
                      f"cut at {4 * budget_s:.0f} s): {plies} plies, slowest worker {dt:.1f} s"}


def _cpu_worker(args):
    cfg = make_cfg(args.sims, args.upper, args.board)
    weights = os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")
    r = cpu_baseline(cfg, weights if cfg.board_size == 11 else None, budget_s=args.cpu_worker, game_id=args.seed, whole_game=True)
    print(json.dumps({"plies": r["plies"], "dt": r["dt"], "complete": r["complete"]}), flush=True)


class PowerSampler(object):
    """The board's own power sensor and shader clock (amdgpu hwmon under the device's PCI node, read at ~10 Hz by a thread of this
    process) over the timed region: DESIGN's "the forward sits at the power wall" as a sensor reading on the bench line instead of
    an inference from counters.  Reads sysfs only; a box without the sensor gives roofline.power = null."""

    def __init__(self, props):
        import glob
        self.rows, self._stop, self._th, self.paths = [], None, None, None
        try:
            bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
            hw = sorted(glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf))
            if hw:
                pw = [q for q in (hw[0] + "/power1_average", hw[0] + "/power1_input") if os.path.exists(q)]
                if pw:
                    self.paths = {"power": pw[0], "cap": hw[0] + "/power1_cap", "sclk": hw[0] + "/freq1_input"}
        except Exception:
            self.paths = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        if self.paths is None:
            return
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                self.rows.append((self._read(self.paths["power"]), self._read(self.paths["sclk"])))
                self._stop.wait(0.1)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        pw = sorted(r[0] / 1e6 for r in self.rows if r[0] is not None)
        ck = [r[1] / 1e9 for r in self.rows if r[1] is not None]
        cap = self._read(self.paths["cap"])
        if not pw:
            return None
        return {"cap_w": cap / 1e6 if cap else None, "mean_w": sum(pw) / len(pw), "p50_w": pw[len(pw) // 2], "max_w": pw[-1],
                "frac_of_cap": (sum(pw) / len(pw)) / (cap / 1e6) if cap else None,
                "sclk_ghz_mean": sum(ck) / len(ck) if ck else None, "samples": len(pw),
                "source": "amdgpu hwmon %s + freq1_input, 10 Hz over the timed region (rank 0's device)" % os.path.basename(self.paths["power"])}


def copy_bandwidth_gbs(dev, mib=1024, reps=10):
    """Measured device copy bandwidth (read + write bytes / time): the achievable-HBM denominator SURVEY §8d asks for."""
    import torch
    a = torch.empty(mib << 18, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    return 2.0 * a.numel() * 4 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def _short_kernel_name(name):
    import re
    name = re.sub(r"\(anonymous namespace\)::", "", name.strip())
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).replace(" >", ">")[:96]


def live_pmc(G, ticks=400, last=200, timeout_s=180, probe="probe_tick_min.py", extra_env=None, nominal_ghz=2.4):
    """Counters of the forward's kernels, collected now: rocprofv3 --kernel-trace --pmc <set> (kernel trace only, one pass per set:
    MI355X_MICROARCH.md) around tools/<probe> in child processes, mean of each kernel's last `last` dispatches.
      pass 1 / 2  FETCH_SIZE, WRITE_SIZE         -> HBM bytes per forward = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, gfx950 correction
      pass 3      GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES + the pass's own kernel durations -> per kernel: effective clock
                  (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA-busy share (busy cycles / (1024 SIMDs x active cycles)), MFMA FLOP/s
                  issued (busy cycles / 32 per v_mfma_f32_32x32x16 x 32,768 FLOP / duration)
    -> dict(traffic=bytes or None, per_kernel=[...], sustained_clock_ghz, mfma_busy) or None if rocprofv3 is missing / a pass fails
    (the caller then keeps the committed profile's figure)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    # not inside another profiler session (bench.py itself run under rocprofv3 / rocprof-sys): no nested tool registration
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCPROFSYS")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="af_pmc_", dir="/tmp")
    per_kernel, durs = {}, {}
    try:
        for ctrs in (("FETCH_SIZE",), ("WRITE_SIZE",), ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES")):
            d = os.path.join(tmp, ctrs[0])
            env = dict(os.environ, TICKS=str(ticks), G=str(G), TMPDIR="/tmp")
            env.update(extra_env or {})
            subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + list(ctrs) + ["-d", d, "-o", "p", "--", sys.executable,
                            os.path.join(REPO, "tools", probe)], cwd="/tmp", env=env, timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if not dbs:
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            name_col = "kernel_name" if "kernel_name" in cols else "name"
            vals = {}
            for k, c, v in cur.execute(f"select {name_col}, counter_name, value from counters_collection order by dispatch_id"):
                if c in ctrs:
                    vals.setdefault((k, c), []).append(v)
            if not vals:
                return None
            for (k, c), vs in vals.items():
                per_kernel.setdefault(k, {})[c] = (sum(vs[-last:]) / len(vs[-last:]), len(vs))
            if len(ctrs) > 1:                          # durations of the same (serialised, profiled) pass
                kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
                kname = "name" if "name" in kcols else "kernel_name"
                for k, a, b in cur.execute(f"select {kname}, start, end from kernels order by start"):
                    durs.setdefault(k, []).append(b - a)
        tick = [k for k in per_kernel if "af_tick_kernel" in k]
        if not tick or "FETCH_SIZE" not in per_kernel[tick[0]]:
            return None
        n_tick = per_kernel[tick[0]]["FETCH_SIZE"][1]
        total, rows, act, busy, ns, ns_all = 0.0, [], 0.0, 0.0, 0.0, 0.0
        for k, c in per_kernel.items():
            if k in tick or "af_pack" in k or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or "af_" not in k:
                continue                               # (torch fill / copy kernels of the probe's set-up)
            per_fwd = round(c["FETCH_SIZE"][1] / n_tick)
            total += per_fwd * (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0
            if "GRBM_GUI_ACTIVE" in c and k in durs:
                dn = durs[k][-last:]
                dur = sum(dn) / len(dn)
                a, m = c["GRBM_GUI_ACTIVE"][0] / 8.0, c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0.0, 0))[0]
                # GRBM_GUI_ACTIVE also counts the dispatch's ramp and drain outside the kernel's own timestamps: below ~50 us
                # active / duration is not a clock (it reads > 2.4 GHz), so short kernels carry no clock and stay out of the mean
                long_enough = dur >= 50e3
                rows.append({"kernel": _short_kernel_name(k), "launches_per_forward": per_fwd, "us": dur / 1e3,
                             "ghz": (a / dur) if long_enough else None,
                             "mfma_busy": (m / (1024.0 * a)) if long_enough else None, "mfma_issued_tflops": m * 1024.0 / dur / 1e3,
                             "hbm_mb": (2.0 * c["FETCH_SIZE"][0] + c["WRITE_SIZE"][0]) * 1024.0 / 1e6})
                if long_enough:
                    act, busy, ns = act + per_fwd * a, busy + per_fwd * m, ns + per_fwd * dur
                ns_all += per_fwd * dur
        out = {"traffic": int(total) if total > 0 else None, "per_kernel": sorted(rows, key=lambda r: -r["us"]), "nominal_ghz": nominal_ghz}
        if ns > 0:
            out.update({"sustained_clock_ghz": act / ns, "mfma_busy": busy / (1024.0 * act),
                        "profiled_forward_us": ns_all / 1e3, "clocked_share_of_forward": ns / ns_all})
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


FP32_LEG_MODE = 1      # the fp32-MFMA path (Winograd register ring): 3.48 ms per 4096 positions vs 1.42 (tools/probe_arith_width.py, profiles/r6_02)

EXTRA_LEGS = {
    # BASELINE.json configs[3]: 15x15, 800 sims/move (cap 942 = the reference's 642 - 500 head-room), 4096 games
    "config4": ["--board", "15", "--sims", "800", "--upper", "942", "--games", "4096", "--age-plies", "24", "--warmup", "3", "--steps", "6"],
    # BASELINE.json configs[4]: 11x11, 8-block x 128 residual net in bf16, 8192 games
    "config5": ["--net", "deep-bf16", "--games", "8192", "--age-plies", "12", "--warmup", "3", "--steps", "8"],
    # the headline workload once more with the evaluation memo on (same trees, fewer forwards): reported beside the headline, never as it
    "config2_memo": ["--eval-memo", "22:5", "--warmup", "5", "--steps", "20"],
    # the headline workload on arithmetic of the reference's own width: the fp32-MFMA convolution path (24 mantissa bits, 157.3
    # TFLOP/s matrix peak) instead of the 22-bit fp16 split-operand scheme — what the split buys, on the same line
    "config2_fp32mfma": ["--conv-mode", str(FP32_LEG_MODE), "--age-plies", "12", "--warmup", "2", "--steps", "4"],
}


PROCESS_T0 = time.time()


def extra_config_legs(timeout_s=240, wall_budget_s=600.0):
    """Short steady-state legs of the two other single-GPU configurations BASELINE.json names, run as child processes of this
    same file after the default workload (so the driver's one `python bench.py` sees them): flat keys configN_* on the JSON
    line plus the children's own lines under "extra_configs".  A leg that fails or times out is reported as an error string."""
    import subprocess
    flat, full = {}, {}
    for name, leg in EXTRA_LEGS.items():
        # the whole default run is meant to finish within minutes also on a box that pages the image in for the first time (every
        # child start then costs tens of seconds): a leg that would start past the budget is skipped and says so
        if time.time() - PROCESS_T0 > wall_budget_s - 60.0:
            flat[name + "_error"] = "skipped: %.0f s of the run's %.0f-s wall budget were used before this leg" % (time.time() - PROCESS_T0, wall_budget_s)
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-pmc", "--no-extra-configs"] + leg
        t0 = time.time()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s,
                               env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
            line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                flat[name + "_error"] = "rc %d: %s" % (r.returncode, r.stderr.decode()[-300:])
                continue
            j = json.loads(line[-1])
        except subprocess.TimeoutExpired:
            flat[name + "_error"] = "timed out after %d s" % timeout_s
            continue
        rf = j["roofline"]
        if j["value"] is None:                      # no episode finished inside the leg: say so instead of quoting it as the metric
            flat[name + "_steady_state"] = False
            j["value"] = j.get("opening_phase_moves_per_s")
        flat.update({name + "_moves_per_s": j["value"], name + "_net_ms": rf["ms_per_launch"],
                     name + "_tick_ms": j["tree_roofline"]["ms_per_launch"], name + "_mfma_frac": rf["frac"],
                     name + "_net_tflops": rf["achieved"], name + "_mfma_peak_tflops": rf["peak"],
                     name + "_episodes_finished": j["config"]["episodes_finished_in_timed_region"],
                     name + "_steps": j["steps"], name + "_ms_per_step": j["ms_per_step"], name + "_wall_s": time.time() - t0})
        if "mfma_issued_frac" in rf:
            flat[name + "_mfma_issued_frac"] = rf["mfma_issued_frac"]
        for k in ("dv_max_vs_torch_fp32", "dp_max_vs_torch_fp32", "operand_mantissa_bits"):
            if k in rf:
                flat[name + "_" + k] = rf[k]
        if j["config"].get("eval_memo"):
            flat[name + "_hits_per_simulation"] = j["config"]["eval_memo"]["hits_per_simulation"]
        full[name] = {"cmd": " ".join(["python", "bench.py"] + cmd[2:]), "metric": j["metric"], "value": j["value"], "dtype": j["dtype"],
                      "config": j["config"], "roofline": {k: rf[k] for k in ("kernel", "achieved", "peak", "frac", "ms_per_launch")},
                      "time_split": j["time_split"]}
    # BASELINE configs[0] through the drop-in API: one eval-mode game of the self_play.py:79-106 loop, Player(pv_fn=net.eval) on the
    # device / HIP-graph path (alphafive_amd/self_play.py) — next to cpu_baseline.config1_eval_mode_value (C port, 1 core)
    try:
        r = subprocess.run([sys.executable, "-m", "alphafive_amd.self_play", "--weights-npz",
                            os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")], cwd=REPO, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=120)
        last = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("game finished")]
        if r.returncode == 0 and last:
            tok = last[-1].split()                  # "game finished: N plies in T s = X moves/s"
            flat["config1_player_moves_per_s"] = float(tok[-2])
            flat["config1_player_plies"] = int(tok[2])
        else:
            flat["config1_player_error"] = "rc %d: %s" % (r.returncode, r.stderr.decode()[-200:])
    except subprocess.TimeoutExpired:
        flat["config1_player_error"] = "timed out"
    flat["extra_configs"] = full
    return flat


def spawn_ranks(n, timeout_s=None):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves — the same command line once
    per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in the environment, i.e. what
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would have set.  The children inherit stdout, and only
    rank 0 prints the JSON line.  A rank that fails takes the others down with it (no hang on a half-formed group)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AF_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True))
    def _stop(signum, frame):                     # the launcher being stopped (a driver's timeout) must not leave ranks behind
        raise KeyboardInterrupt
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, _stop)
    rc, t0 = 0, time.time()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout_s and time.time() - t0 > timeout_s):
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)      # exactly the process groups started above
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
    if rc != 0:
        raise SystemExit(rc if rc > 0 else 1)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=500)
    ap.add_argument("--upper", type=int, default=642)
    ap.add_argument("--board", type=int, default=11, help="board size (15 with --sims 800 --upper 942 = BASELINE configs[3])")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--poll", type=int, default=16, help="ticks per progress poll (= ticks per HIP-graph replay with --loop graph)")
    ap.add_argument("--loop", default="graph", choices=["graph", "eager"],
                    help="graph (default): the steady-state loop replays --poll x (tick + forward) as ONE HIP graph and reads the "
                         "progress words one replay late, so the host never drains the device; one eager chunk per step carries the "
                         "HIP events that time the tick kernel and the forward.  eager: every launch issued from Python (r3's loop; A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="take roofline.traffic from profiles/ instead of two rocprofv3 counter passes in this run")
    ap.add_argument("--age-plies", type=int, default=0,
                    help="before the warm-up steps, age the games this many plies at --age-sims simulations per move (untimed): the "
                         "games desynchronise and reach the mixed-phase population of the steady state ~40x faster than full-budget "
                         "plies would; the warm-up steps then rebuild full-budget trees.  Used by the short extra-config legs.")
    ap.add_argument("--age-sims", type=int, default=16)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short steady-state legs of BASELINE configs[3] (15x15, 800 sims) and configs[4] (8-block bf16 net, "
                         "8192 games) that the default N=1 run appends as flat keys config4_* / config5_*")
    ap.add_argument("--eval-memo", default="", metavar="LOG2_BUCKETS:MAX_STONES",
                    help="share evaluations between the games of a rank (include/af_engine.h ABI v5; e.g. 22:5).  Off for the headline: "
                         "the default run measures it in a leg of its own (config2_memo_*)")
    ap.add_argument("--pipe-values", action="store_true",
                    help="W / Q in fp64: the arithmetic of main.py's pipe-fed workers (networkAPI.py:72); default = the pv_fn path")
    ap.add_argument("--conv-mode", type=int, default=5, choices=[1, 5],
                    help="af_net_tune(0, .): 5 (default) = fp16 split-operand implicit GEMM (22 mantissa bits, 3 fp16 MFMA products per "
                         "MAC); 1 = the fp32-MFMA Winograd path of af_net.hip (24 bits, the reference's width).  The default run measures "
                         "mode 1 in a leg of its own (config2_fp32mfma_*)")
    ap.add_argument("--net", default="hip", choices=["hip", "torch", "deep-bf16", "deep-bf16-torch"],
                    help="hip (default): the hand-written kernels, fails without libaf_net.so; torch: PyTorch-ROCm ops "
                         "(reference only); deep-bf16 = BASELINE configs[4]: 8-block width-128 net in bf16 (use with --games 8192)")
    args = ap.parse_args()
    if args.cpu_worker > 0:
        return _cpu_worker(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N` (the way the driver starts --gpus 1): be the launcher — one rank per
        # GPU as child processes of this file (main.py:50-55 forks its own workers too); rank 0's line is the only output
        return spawn_ranks(args.gpus)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (python -m torch.distributed.run --nproc-per-node %d "
                         "bench.py --gpus %d) or plain `python bench.py --gpus %d`, which spawns its own ranks"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    # AF_BENCH_SHARE_GPU=1 (test hook): all ranks share device 0 and talk over gloo, so the N>1 code path
    # can be exercised end to end on a 1-GPU box; the driver's runs use one GPU per rank over RCCL.
    share = os.environ.get("AF_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    comm_dev = torch.device("cpu") if share else dev

    from alphafive_amd.engine import SelfPlayEngine
    from alphafive_amd.network import ResNet
    from alphafive_amd import dist as afdist

    cfg = make_cfg(args.sims, args.upper, args.board)
    G, C = args.games, cfg.board_size ** 2
    weights = os.path.join(REPO, "tests", "golden", "alphaFive-6960.weights.npz")
    net = ResNet(cfg.board_size, device=dev, seed=0)
    if cfg.board_size == 11:
        net.load_npz(weights)                     # other sizes: random init of the same architecture (no checkpoint exists)
    if world > 1:
        afdist.broadcast_weights(net, src=0)      # one 3 MB broadcast, as a weight update would do
    deep = None
    if args.net.startswith("deep-bf16"):
        from alphafive_amd.network_deep import DeepResNet
        deep = DeepResNet(cfg.board_size, blocks=8, width=128, device=dev)
        pv = deep.select_backend("torch" if args.net.endswith("torch") else "hip", G)   # "hip" raises without libaf_tower.so
    else:
        pv = net.select_backend(args.net)
        if args.conv_mode != 5:
            if args.net != "hip" or cfg.board_size not in (11, 15):
                raise SystemExit("--conv-mode selects a path of the hand-written 11x11 / 15x15 forward (--net hip)")
            from alphafive_amd import net_hip
            net_hip.tune(0, args.conv_mode)
    memo = None
    if args.eval_memo:
        lb, ms = (int(x) for x in args.eval_memo.split(":"))
        memo = dict(log2_buckets=lb, max_stones=ms)
    sp = SelfPlayEngine(cfg, G, pv, device=local, seed=args.seed, first_game_id=rank * G, value_f64=args.pipe_values, eval_memo=memo)
    stream = torch.cuda.current_stream(dev).cuda_stream

    ev_tick, ev_net = [], []
    timing = {"on": False}
    gathered = {"episodes": 0, "plies": 0, "by_rank": [0] * world, "max_gid": -1}

    def one_tick():
        if timing["on"]:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            sp.engine.tick(sp.policy.data_ptr(), sp.value.data_ptr(), sp.planes.data_ptr(), stream)
            e1.record()
            p, v = pv(sp.planes)
            if p.data_ptr() != sp.policy.data_ptr():
                sp.policy.copy_(p)
                sp.value.copy_(v)
            e2.record()
            if memo:
                sp.engine.memo_insert(sp.policy.data_ptr(), sp.value.data_ptr(), stream)
            ev_tick.append((e0, e1))
            ev_net.append((e1, e2))
            sp.ticks += 1
        else:
            sp.tick()

    EP_CAP = 512                                      # episodes per hand-off (~G/26 finish per step; the rest waits)
    use_graph = args.loop == "graph"
    SAMPLE = 8                                        # eager, event-timed ticks per step in graph mode (~2 % of a step's ticks)

    stepno = {"n": 0}

    def run_step(target):
        sample = use_graph and timing["on"]
        # where in the step the eager sample sits rotates from step to step (replay 0, 5, 10, 15, 0, ... of ~26), so that the
        # tick-kernel duration the forward's in-graph time is derived from is not always measured right after a hand-off
        at, k = (stepno["n"] * 5) % 20, 0
        stepno["n"] += 1
        while True:
            if use_graph:
                if sample and k == at and 3 * args.poll <= 64:
                    # one replay per step is the STAMPED capture of the same ticks: three device-clock stamps per tick (before the tree
                    # kernel, between the two, after the forward) = each kernel's duration as it runs inside the graph
                    sp.run_ticks_graph(args.poll, stamped=True)
                else:
                    sp.run_ticks_graph(args.poll, timed=timing["on"])     # returns at once
                if sample and k >= at:
                    # the sample that times the two kernels: eager launches bracketed by HIP events on the launch stream, inside
                    # the timed region, issued while the replay above keeps the device busy (no launch latency inside the
                    # brackets); everything else of the step runs from the graph
                    for _ in range(SAMPLE):
                        one_tick()
                    sample = False
                k += 1
                plies, _ = sp.progress_lagged()       # as of the previous replay: the device stays busy with the newest one
                if plies >= target and sample:        # a short step: it still carries its sample
                    for _ in range(SAMPLE):
                        one_tick()
                    sample = False
            else:
                for _ in range(args.poll):
                    one_tick()
                plies, _ = sp.progress()
            if plies >= target:
                break
        sp.check()
        hand_off()

    posted = {"buf": None}

    gat = None
    if world > 1:
        # finished episodes -> rank 0 only: sizes ride a tiny all-gather issued with the pack, the used prefixes follow one step
        # later as ONE gather towards rank 0 (RCCL grouped send/recv over xGMI); nothing here waits for the device
        gat = afdist.EpisodeGather(world, rank, comm_dev, EP_CAP, 2 * sp.engine.KW2 + 2 * C + 2, games_per_rank=G)

    handoff = {"s": 0.0, "n": 0}

    def collect(last=False):
        # header-only accounting: no per-episode host work on any rank between sp.check() and the next tick
        # N = 1: the pack kernels wrote the episodes into pinned host memory one step ago
        if world == 1:
            if posted["buf"] is None:
                return
            buf = sp.collect_episodes(EP_CAP, unpack=False)
            posted["buf"] = None
            n_eps, n_plies = (int(buf[0]), int(buf[1])) if buf is not None else (0, 0)
        else:
            got = gat.flush(unpack=False) if last else gat.collect(unpack=False)
            n_eps, n_plies = sum(p.n for p in got), sum(p.plies for p in got)
            for p in got:                             # (rank 0 only; headers only: who sent what, and the largest global game id)
                gathered["by_rank"][p.rank] += p.n
                if p.n:
                    gathered["max_gid"] = max(gathered["max_gid"], p.rank * G + int(p.buf[4:4 + 4 * p.n:4].max()))
        if rank == 0:
            gathered["episodes"] += n_eps
            gathered["plies"] += n_plies

    def hand_off():
        # collect what was posted earlier (its kernels retired long ago: nothing waits), then post this step's
        # episodes behind the ticks already queued — the hand-off never sits on the tick path
        h0 = time.perf_counter()
        collect()
        if world == 1:
            posted["buf"] = sp.post_episodes(EP_CAP)
        else:
            gat.post(sp.post_episodes_device(EP_CAP))
        if timing["on"]:
            handoff["s"] += time.perf_counter() - h0
            handoff["n"] += 1

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    target = 0
    if args.age_plies > 0:
        sp.engine.set_simulations(args.age_sims, args.age_sims + 8)
        for _ in range(args.age_plies):
            target += G
            run_step(target)
        sp.engine.set_simulations(args.sims, args.upper)
        target = sp.progress()[0]                     # (the lagged poll lets a phase overshoot its last target: start from the truth)
    for _ in range(args.warmup):
        target += G
        run_step(target)
    barrier()
    target = sp.progress()[0]
    ct0 = sp.counters()
    ms0 = sp.engine.memo_stats(stream) if memo else None
    p0 = sp.progress()[0]
    sp.engine.tick_histogram(stream, reset=True)
    timing["on"] = True
    ticks0 = sp.ticks
    gathered_at_t0 = gathered["episodes"]
    power = PowerSampler(torch.cuda.get_device_properties(dev)) if rank == 0 else None
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        target += G
        run_step(target)
    collect(last=True)                                # the last steps' episodes
    barrier()
    elapsed = time.perf_counter() - t0
    power_report = power.stop() if power is not None else None
    gathered_in_region = gathered["episodes"] - gathered_at_t0
    timing["on"] = False
    ticks_timed = sp.ticks - ticks0
    ct1 = sp.counters()
    memo_report = None                               # rank 0's engine, timed region only
    if memo:
        ms1 = sp.engine.memo_stats(stream)
        dm = {k: ms1[k] - ms0[k] for k in ("launches", "probes", "hits", "inserts", "replaced")}
        memo_report = dict(memo, entries=ms1["entries"], bytes=ms1["entries"] * (584 if cfg.board_size ** 2 <= 128 else 1160), **dm,
                           hits_per_simulation=dm["hits"] / max(1, ct1["sims"] - ct0["sims"]),
                           note="evaluations shared between the games of a rank: a leaf another game had evaluated before is expanded "
                                "at once from the stored bits; trees are bit-identical to the run without it (tests/test_gpu_eval_memo.py)")
    hist = sp.engine.tick_histogram(stream)
    plies = sp.progress()[0] - p0

    tot = torch.tensor([float(plies)], device=comm_dev, dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=comm_dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_plies, t = float(tot.item()), float(tmax.item())
    eps_all = torch.tensor([float(ct1["episodes"] - ct0["episodes"])], device=comm_dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(eps_all, op=dist.ReduceOp.SUM)
    steady = eps_all.item() > 0
    # what the process group really was (self-verifying N>1 line): an all-reduce of 1 over it, its backend, every rank's device
    # and per-rank counts, so that "RCCL saw N ranks on N different GPUs" can be read off the line instead of the environment
    ones = torch.ones(1, device=comm_dev, dtype=torch.float64)
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "device": "cuda:%d" % local, "name": props.name,
          "uuid": str(getattr(props, "uuid", "")),
          "pci_bus_id": "%s:%s:%s" % (getattr(props, "pci_domain_id", "?"), getattr(props, "pci_bus_id", "?"), getattr(props, "pci_device_id", "?")),
          "episodes_finished": int(ct1["episodes"] - ct0["episodes"]), "episodes_total": int(ct1["episodes"]), "plies": int(plies),
          "elapsed_s": elapsed}
    ranks_info = [me]
    if world > 1:
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    ranks_seen = int(ones.item())
    backend = dist.get_backend() if world > 1 else None
    # untimed: drain the episodes still waiting on the devices (more than EP_CAP may finish in a step, and the gather runs one
    # step behind), so that the line can state "rank 0 holds every episode the ranks finished": cumulative counts since engine
    # creation on both sides (the engines' device counters vs the headers rank 0 received)
    fin_total = torch.tensor([float(ct1["episodes"])], device=comm_dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(fin_total, op=dist.ReduceOp.SUM)
    got_total = torch.zeros(1, device=comm_dev, dtype=torch.float64)
    for _ in range(16):
        got_total[0] = float(gathered["episodes"])
        if world > 1:
            dist.broadcast(got_total, src=0)
        if got_total.item() >= fin_total.item():
            break
        hand_off()
        collect(last=True)
    gathered_total, finished_total = int(got_total.item()), int(fin_total.item())

    if rank == 0:
        arith = None
        if deep is None and args.net == "hip":
            # arithmetic width on the line: the live leaf batch of the last tick through the kernels once more
            # (same bits: the forward is deterministic) and through ResNet.eval_torch = PyTorch-ROCm fp32 ops, product side, no oracle
            pl, vl = (x_.clone() for x_ in pv(sp.planes))
            pt_, vt_ = net.eval_torch(sp.planes)
            arith = {"dv_max_vs_torch_fp32": float((vl - vt_).abs().max().item()), "dp_max_vs_torch_fp32": float((pl - pt_).abs().max().item()),
                     "dv_mean_vs_torch_fp32": float((vl - vt_).abs().mean().item()), "positions": int(G),
                     "operand_mantissa_bits": 22 if args.conv_mode == 5 and cfg.board_size in (11, 15) else 24}
            del pl, vl, pt_, vt_
        d = {k: ct1[k] - ct0[k] for k in ct1}
        n_ticks = ticks_timed                       # every tick of the timed region (graph replays + the event-timed sample)
        tick_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_tick]))
        net_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_net]))
        net_ms_eager = net_ms
        graph_tick_ms = None
        stamps = sp.read_stamps() if use_graph else np.zeros((0, 2))
        derived_net_ms = stamped_tick_ms = None
        if use_graph and sp.replay_events:
            # what the timed region actually runs: the replays, each bracketed by HIP events on its stream.  One tick of a replay =
            # af_tick_kernel + the forward; the tick kernel alone is a single launch whose eager, event-timed samples are what it
            # takes inside the graph too, so the forward inside the graph = replay time per tick - tick kernel.  (The eager samples
            # of the FORWARD carry the cost of its fork / join with the value branch's side stream as stream operations — a box-
            # dependent 0-60 us that a graph edge does not have — and are kept as roofline.ms_per_launch_eager_sample.)
            graph_tick_ms = float(np.sum([a.elapsed_time(b) for _, a, b in sp.replay_events]) / np.sum([n_ for n_, _, _ in sp.replay_events]))
            derived_net_ms = graph_tick_ms - tick_ms
            net_ms = derived_net_ms
            if len(stamps):
                # r6: MEASURED inside the graph — device-clock stamps (af_engine_stamp, 10-ns ticks) around the forward
                # of every tick of one replay per step; each interval carries the launch boundaries of the two stamp kernels around
                # it (~1.5 us each), so it is an upper bound; the derived figure (replay per tick - eager tick kernel) stays beside it
                stamped_tick_ms, net_ms = float(stamps[:, 0].mean()), float(stamps[:, 1].mean())
        flop_pos = FLOP_PER_POSITION if cfg.board_size == 11 else net.flops_per_position()
        roof = net.roofline_info(pv)
        peak, dtype_name = roof.get("peak_tflops", PEAK_FP32_MFMA_TFLOPS), "f32"
        if roof["backend"].startswith("hip") and "f16" in roof["backend"]:
            dtype_name = "f32 (fp16x2 split operands: 3 fp16 MFMA products per MAC, fp32 accumulate)"
        elif roof["backend"].startswith("hip"):
            dtype_name = "f32 (fp32 MFMA operands, fp32 accumulate)"
        if deep is not None:
            flop_pos, peak, dtype_name = deep.flops_per_position(), 2500.0, "bf16"   # dense bf16 MFMA peak
        net_tflops = G * flop_pos / (net_ms * 1e-3) / 1e12
        tree_gbs = tree_bytes(d, C) / n_ticks / (tick_ms * 1e-3) / 1e9
        # HBM traffic per launch: NOT measured live (PMC collection needs its own rocprofv3 passes); taken from the
        # committed counter profile of THIS round's kernels when this run is the profiled workload, else null
        traffic_net = traffic_tick = None
        traffic_src = None
        pmc = None
        tp = next((q for q in (os.path.join(REPO, "profiles", "r%d_pmc_hbm_traffic.json" % r) for r in (6, 5, 4, 3)) if os.path.exists(q)),
                  os.path.join(REPO, "profiles", "r3_pmc_hbm_traffic.json"))
        tp_name = "profiles/" + os.path.basename(tp)
        if os.path.exists(tp) and deep is None and roof["backend"].startswith("hip") and args.conv_mode == 5:
            with open(tp) as f:
                prof = json.load(f)
            if prof["workload"] == {"games": G, "board_size": cfg.board_size}:
                traffic_net = prof["net_forward_bytes_per_launch"]["corrected"]
                traffic_tick = prof["tick_kernel_bytes_per_launch"]["corrected"]
                traffic_src = tp_name + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                # ... and, unless --no-pmc, collected again in THIS run: two short rocprofv3 counter passes over the same kernels
                # (child processes, after the timed region; the net's traffic does not depend on the game phase, the tick
                # kernel's does, so tree_roofline keeps the steady-state figure of the file)
                live = None if (args.no_pmc or world > 1) else live_pmc(G)
                if live is not None and live.get("traffic"):
                    pmc = live
                    traffic_net = live["traffic"]
                    traffic_src = ("this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                                   "tools/probe_tick_min.py, bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the forward's "
                                   "kernels, mean of the last 200 dispatches; %s has %.3f GB"
                                   % (tp_name, prof["net_forward_bytes_per_launch"]["corrected"] / 1e9))
        if deep is not None:
            roof = ({"backend": "torch-rocm bf16 (MIOpen/hipBLASLt), 8 residual blocks x 128",
                     "kernel": "deep net forward (PyTorch-ROCm ops, whole forward timed)"} if args.net.endswith("torch") else
                    {"backend": "hip (af_tower_bf16.hip: bf16 MFMA stem + implicit-GEMM tower, weight-stationary, LDS-DMA staging + heads 1x1 convs + MFMA dense layers / softmax)",
                     "kernel": "deep net forward = af_tower_stem + 16x af_tower_conv + af_tower_heads + af_tower_dense (whole forward timed; af_tower_conv carries 99 % of the FLOPs)"})
        copy_gbs = copy_bandwidth_gbs(dev)
        out = {
            "metric": "self-play moves/sec (%dx%d, %d sims/move)" % (cfg.board_size, cfg.board_size, args.sims),
            "value": (total_plies / t) if steady else None, "unit": "moves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[4]: {G} concurrent 11x11 games per GPU, {args.sims} sims/move (cap {args.upper}), "
                                    f"training-mode MCTS, 8-block x 128 residual net in bf16 (random init), batched leaf eval")
                       if deep is not None else
                       (f"BASELINE configs[1]: {G} concurrent 11x11 games per GPU, {args.sims} sims/move "
                        f"(cap {args.upper}), training-mode MCTS, alphaFive-6960 net ("
                        + ("fp32-class: fp16 split operands, fp32 accumulate" if args.conv_mode == 5 else "fp32 MFMA operands, af_net_tune(0, %d)" % args.conv_mode)
                        + "; within 1e-5 of the fp64 restatement), batched leaf eval")
                       if cfg.board_size == 11 else
                       (f"BASELINE configs[3]-style: {G} concurrent {cfg.board_size}x{cfg.board_size} games per GPU, "
                        f"{args.sims} sims/move (cap {args.upper}), random-init net of the same architecture, fp32"),
                       "games_per_gpu": G, "sims_per_move": args.sims, "tree_value_dtype": "f64 (pipe path)" if args.pipe_values else "f32 (pv_fn path)",
                       "net_backend": roof["backend"],
                       "step": "ticks until the batch commits G more plies", "ticks_timed_rank0": n_ticks,
                       "loop": ("HIP graph: %d x (af_tick_kernel + af_net_forward) + progress copy per replay, progress read one replay "
                                "late; %d eager event-timed ticks per step" % (args.poll, SAMPLE)) if use_graph else
                               ("eager: every launch issued from Python, progress poll every %d ticks" % args.poll),
                       "ticks_event_timed_rank0": len(ev_tick),
                       "sims_per_ply_rank0": d["sims"] / max(1, d["plies"]),
                       "selects_per_sim": d["selects"] / max(1, d["sims"]),
                       "terminal_frac": d["terminals"] / max(1, d["sims"]),
                       "episodes_gathered": gathered_in_region, "episodes_finished_in_timed_region": int(eps_all.item()),
                       "eval_memo": memo_report,
                       # self-verification of the N>1 line: the process group as it really was, and rank 0's receipt of every
                       # episode any rank finished since its engine was created (after an untimed drain of what was still queued)
                       "ranks_seen": ranks_seen, "backend": backend if world > 1 else "none (single process)",
                       "launcher": "bench.py spawn_ranks" if os.environ.get("AF_BENCH_SPAWNED") == "1" else
                                   ("torch.distributed.run / external" if world > 1 else "single process"),
                       "devices": ["%s %s pci %s uuid %s" % (r["device"], r["name"], r["pci_bus_id"], r["uuid"][:13]) for r in ranks_info],
                       "distinct_devices": len({(r["uuid"] or r["pci_bus_id"]) for r in ranks_info}),
                       "per_rank": [{"rank": r["rank"], "episodes_finished": r["episodes_finished"], "plies": r["plies"],
                                     "elapsed_s": r["elapsed_s"]} for r in ranks_info],
                       "episodes_gathered_total": gathered_total, "episodes_finished_total_all_ranks": finished_total,
                       "gathered_equals_finished": gathered_total == finished_total,
                       # N > 1: whose episodes rank 0 holds, read off the packed headers it received (source rank r's games are
                       # the global ids r*G .. r*G + G-1), against every rank's own device counter
                       "episodes_gathered_by_source_rank": gathered["by_rank"] if world > 1 else None,
                       "episodes_finished_by_rank_total": [r["episodes_total"] for r in ranks_info] if world > 1 else None,
                       "max_global_game_id_gathered": gathered["max_gid"] if world > 1 else None,
                       "gather_ring_allocations": gat.allocations if gat is not None else None,
                       # rank 0's host time in the per-step hand-off (collect + pack launch + post), wall clock, inside the timed
                       # region: what the other ranks would wait for at max-over-ranks timing
                       "rank0_handoff_ms_per_step": 1e3 * handoff["s"] / max(1, handoff["n"])},
            "roofline": {"kernel": roof["kernel"], "bound": "mfma", "achieved": net_tflops,
                         "peak": peak, "unit": "TFLOP/s", "frac": net_tflops / peak,
                         "traffic": traffic_net, "traffic_source": traffic_src, "ms_per_launch": net_ms,
                         "ms_per_launch_eager_sample": net_ms_eager,
                         "ms_per_launch_source": ("MEASURED inside the HIP graph: device-clock stamps (af_engine_stamp: s_memrealtime, 10-ns ticks) before and "
                                                  "after the forward of every tick of one stamped replay per step (%d ticks stamped; timing events "
                                                  "cannot be captured on ROCm); includes the launch boundaries of the two stamp kernels (~1.5 us each). "
                                                  "ms_per_launch_derived = graph replay time per tick - the tick kernel's eager event-timed duration"
                                                  % len(stamps)) if len(stamps) else
                                                 ("DERIVED: HIP events around every graph replay / ticks per replay - the tick kernel's eager "
                                                  "event-timed duration") if graph_tick_ms else
                                                 "HIP events around every eager forward of the timed region",
                         "ms_per_launch_derived": derived_net_ms,
                         "flop_per_launch": G * flop_pos,
                         "note": "achieved = algorithmic (direct-convolution) FLOPs per launch / launch time; peak = dense MFMA peak of the "
                                 "operand type the kernel issues (fp16: 2.5 PFLOP/s; fp32: 157.3 TFLOP/s)"},
            "tree_roofline": {"kernel": "af_tick_kernel<%d>" % (2 if C <= 128 else 4), "bound": "hbm", "achieved": tree_gbs,
                              "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": tree_gbs / PEAK_HBM_GBS, "traffic": traffic_tick,
                              "ms_per_launch": tick_ms, "bytes_per_launch": tree_bytes(d, C) / n_ticks,
                              "peak_measured_copy": copy_gbs, "frac_of_measured_copy": tree_gbs / copy_gbs,
                              "note": "SURVEY 8d bound (HBM); PMC (profiles/r1_16) shows the kernel limited by per-game serial latency and fp64 VALU work of the noise generator"},
            "time_split": {"tree_ms_per_tick": tick_ms, "net_ms_per_tick": net_ms,
                           # what a tick costs beyond its two kernels: launch gaps, host polls, the hand-off (<= 5 us)
                           "outside_kernels_us_per_tick": 1e3 * (1e3 * t / max(1, n_ticks) - (graph_tick_ms if graph_tick_ms else tick_ms + net_ms)),
                           "graph_replay_ms_per_tick": graph_tick_ms, "net_ms_per_tick_eager_sample": net_ms_eager,
                           "tree_ms_per_tick_in_graph_stamped": stamped_tick_ms, "ticks_stamped": int(len(stamps)),
                           "tree_ms_max": float(np.max([a.elapsed_time(b) for a, b in ev_tick])),
                           "tree_ms_p50": float(np.median([a.elapsed_time(b) for a, b in ev_tick]))},
            "tick_shape": {"selects_per_game_and_launch_hist": hist["selects"].tolist(),
                           "wave_lifetime_8us_bins": hist["wave_us"].tolist(), "max_wave_us": hist["max_wave_us"],
                           "collector_runs": d["collector_runs"], "collector_slots_scanned": d["collector_scanned"],
                           "yields": d["yields"], "stalls": d["stalls"]},
        }
        if "issued_flop_per_position" in roof and deep is None:
            issued = G * roof["issued_flop_per_position"] / (net_ms * 1e-3) / 1e12
            out["roofline"]["mfma_issued_tflops"] = issued          # MFMA FLOPs actually issued (3 per algorithmic MAC)
            out["roofline"]["mfma_issued_frac"] = issued / peak
            hb = G * roof["algorithmic_bytes_per_position"]
            out["roofline"]["hbm_algorithmic_bytes_per_launch"] = hb
            out["roofline"]["hbm_algorithmic_gbs"] = hb / (net_ms * 1e-3) / 1e9
            out["roofline"]["hbm_frac_of_peak"] = hb / (net_ms * 1e-3) / 1e9 / PEAK_HBM_GBS
        if arith is not None:
            out["roofline"].update(arith)
            out["roofline"]["arithmetic_note"] = ("d*_vs_torch_fp32: this run's live leaf batch, kernels vs ResNet.eval_torch (PyTorch-ROCm fp32 ops); the "
                                                  "tests hold the value to 1e-5 of the fp64 restatement; config2_fp32mfma_* = the same workload on the "
                                                  "fp32-MFMA path (24-bit operands)")
        out["roofline"]["power"] = power_report                 # the board's sensor over the timed region (None without one)
        if pmc is not None and "sustained_clock_ghz" in pmc:
            # what DESIGN argues from the counters, reproducible from this line alone: the chip is power-bound under this load, so the
            # MFMA pipe's ceiling is the clock it sustains, not the nominal 2.4 GHz the 2.5 PFLOP/s peak is quoted at
            r = out["roofline"]
            sust_peak = peak * pmc["sustained_clock_ghz"] / pmc["nominal_ghz"]
            r.update({"sustained_clock_ghz": pmc["sustained_clock_ghz"], "nominal_clock_ghz": pmc["nominal_ghz"],
                      "sustained_peak_tflops": sust_peak, "mfma_busy": pmc["mfma_busy"],
                      "frac_of_sustained_peak": r["achieved"] / sust_peak,
                      "per_kernel": pmc["per_kernel"], "profiled_forward_us": pmc["profiled_forward_us"],
                      "clocked_share_of_forward": pmc["clocked_share_of_forward"],
                      "clock_source": "this run: rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES over "
                                      "tools/probe_tick_min.py (kernels serialised by the profiler; value branch not overlapped): clock = "
                                      "GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, time-weighted over the forward's kernels of >= 50 us "
                                      "(clocked_share_of_forward of its time; shorter dispatches: the counter also covers ramp and drain); "
                                      "mfma_busy = busy cycles / (1024 SIMDs x active cycles) over the same kernels"})
            if "mfma_issued_tflops" in r:
                r["mfma_issued_frac_of_sustained_peak"] = r["mfma_issued_tflops"] / sust_peak
        if not steady:
            out["invalid"] = ("no episode finished inside the run: this is an opening-phase rate, not the steady-state metric "
                              "(SURVEY 8d); use the defaults (--warmup 8 --steps 20)")
            out["opening_phase_moves_per_s"] = total_plies / t
        if not args.no_cpu_baseline and world == 1:            # reported baseline: rank 0 at N=1 only
            w11 = weights if cfg.board_size == 11 else None
            cb = cpu_baseline(cfg, w11)                                            # the bench's own workload, 1 core
            # (1) config 1 (eval mode, whole game), (2) every usable core on complete episodes, (3) tree only.
            # Flat keys: the driver's parser keeps scalars.
            c1 = cpu_baseline(cfg, w11, training=False, whole_game=True)
            cb.update({"config1_eval_mode_value": c1["value"], "config1_eval_mode_sample": c1["sample"],
                       "tree_only_sims_per_s": cpu_tree_only(cfg)})
            ac = cpu_baseline_all_cores(cfg, w11)
            if ac:
                cb.update({"all_cores_value": ac["value"], "all_cores_cores": ac["cores"], "all_cores_host_cores": ac["host_cores"],
                           "all_cores_sample": ac["sample"]})
            out["cpu_baseline"] = cb
        if args.age_plies > 0:
            out["config"]["aged"] = ("games aged %d plies at %d sims/move before the %d full-budget warm-up steps (untimed)"
                                     % (args.age_plies, args.age_sims, args.warmup))
        default_workload = (world == 1 and deep is None and cfg.board_size == 11 and G == 4096 and args.sims == 500
                            and not args.pipe_values and args.net == "hip")
        if default_workload and not args.no_extra_configs:
            sp.close()
            sp = None
            torch.cuda.empty_cache()
            out.update(extra_config_legs())
        print(json.dumps(out), flush=True)
    if sp is not None:
        sp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

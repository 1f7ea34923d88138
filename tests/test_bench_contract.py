"""The one JSON line `python bench.py` prints, checked on the line the GPU box produced last (committed under profiles/): the
keys the driver's contract names, the roofline / cpu_baseline objects, and the arithmetic that ties them together."""
import glob
import json
import os

import pytest

from conftest import REPO


def _last_line():
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r6_*_bench_driver_command.json")))
    assert files, "no committed bench line of this round"
    with open(files[-1]) as f:
        lines = [ln for ln in f if ln.startswith("{")]
    return json.loads(lines[-1])


def test_bench_line_carries_the_contract():
    d = _last_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("self-play moves/sec (11x11, 500 sims/move)") and d["unit"] == "moves/s"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and d["dtype"].startswith("f32") and "workload" in d["config"] and "model" not in d["config"]
    assert "4096 concurrent 11x11 games" in d["config"]["workload"]
    # value = plies committed / wall time; a step commits one ply per game
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["games_per_gpu"]) < 0.02 * d["config"]["games_per_gpu"]
    assert d["config"]["episodes_finished_in_timed_region"] > 0            # steady state, not the opening phase
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["ms_per_launch"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0.9 < r["traffic"] / r["hbm_algorithmic_bytes_per_launch"] < 1.5   # counters vs algorithmic bytes
    # the dominant kernel cannot take longer than a step allows
    ticks = d["config"]["ticks_timed_rank0"] / d["steps"]
    assert ticks * (r["ms_per_launch"] + d["tree_roofline"]["ms_per_launch"]) < 1.03 * d["ms_per_step"]
    # r4: the line carries what DESIGN argues about the power-bound chip — reproducible from the line alone
    assert 1.5 < r["sustained_clock_ghz"] < 2.45 and r["nominal_clock_ghz"] == 2.4 and 0.3 < r["mfma_busy"] < 0.9
    assert abs(r["sustained_peak_tflops"] - r["peak"] * r["sustained_clock_ghz"] / 2.4) < 1e-6 * r["peak"]
    assert abs(r["frac_of_sustained_peak"] - r["achieved"] / r["sustained_peak_tflops"]) < 1e-9
    pw = r["power"]                                                            # the board's sensor over the timed region
    assert pw is None or (0.5 < pw["frac_of_cap"] <= 1.02 and 1.0 < pw["sclk_ghz_mean"] < 2.5 and pw["samples"] >= 20)
    assert abs(r["mfma_issued_frac_of_sustained_peak"] - r["mfma_issued_tflops"] / r["sustained_peak_tflops"]) < 1e-9
    pk = r["per_kernel"]
    assert len(pk) >= 8 and all(set(("kernel", "us", "ghz", "mfma_busy", "mfma_issued_tflops", "hbm_mb")) <= set(e) for e in pk)
    assert abs(sum(e["us"] * e["launches_per_forward"] for e in pk) - r["profiled_forward_us"]) < 1e-6 * r["profiled_forward_us"]
    assert abs(sum(e["hbm_mb"] * e["launches_per_forward"] for e in pk) * 1e6 - r["traffic"]) < 0.01 * r["traffic"]
    # the forward as it runs inside the graph: replay time per tick - tick kernel; what lies outside the replays is small
    ts = d["time_split"]
    assert abs(ts["graph_replay_ms_per_tick"] - ts["tree_ms_per_tick"] - r.get("ms_per_launch_derived", r["ms_per_launch"])) < 1e-9
    if "ms_per_launch_derived" in r and ts.get("ticks_stamped"):
        # r6: ms_per_launch is MEASURED inside the graph by device-clock stamps (an upper bound: it carries two launch boundaries);
        # the derived figure stays beside it and the two must agree within 3 %
        assert "MEASURED inside the HIP graph" in r["ms_per_launch_source"] and ts["ticks_stamped"] >= 16
        assert 0.0 <= r["ms_per_launch"] / r["ms_per_launch_derived"] - 1.0 < 0.03
        assert abs(ts["tree_ms_per_tick_in_graph_stamped"] / ts["tree_ms_per_tick"] - 1.0) < 0.15
    # r6: arithmetic width on the line
    assert r["operand_mantissa_bits"] == 22 and 0 < r["dv_max_vs_torch_fp32"] < 5e-5 and 0 < r["dp_max_vs_torch_fp32"] < 5e-5
    assert abs(ts["outside_kernels_us_per_tick"]) < 10.0 and "HIP graph" in d["config"]["loop"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["unit"] == "moves/s" and c["value"] > 0


def test_extra_config_legs_are_flat_scalars():
    d = _last_line()
    for name in ("config4", "config5"):
        assert isinstance(d[name + "_moves_per_s"], float) and d[name + "_moves_per_s"] > 0
        assert d[name + "_episodes_finished"] > 0 and d[name + "_net_ms"] > 0 and 0 < d[name + "_mfma_frac"] < 1
    assert "15x15" in d["extra_configs"]["config4"]["metric"] and d["extra_configs"]["config5"]["dtype"] == "bf16"
    # r6: the headline workload on 24-bit operands (fp32 MFMA), beside the 22-bit headline: slower, and no closer to PyTorch fp32
    assert 0 < d["config2_fp32mfma_moves_per_s"] < d["value"] and d["config2_fp32mfma_operand_mantissa_bits"] == 24
    assert d["config2_fp32mfma_mfma_peak_tflops"] == 157.3 and 0 < d["config2_fp32mfma_mfma_frac"] <= 1.0
    assert d["config1_player_moves_per_s"] > 50                              # the single-launch small-batch forward (r5: 42)


@pytest.mark.gpu
def test_a_fresh_bench_run_prints_the_contract_line():
    """the checks above read a committed artefact, so they pass whatever bench.py prints today.  This one runs
    bench.py now (short: 2 timed steps on 512 games, no CPU baseline / counter passes / extra legs) and checks the line it prints."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--games", "512", "--steps", "2", "--warmup", "1", "--age-plies", "30",
                        "--no-cpu-baseline", "--no-pmc", "--no-extra-configs"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "tree_roofline", "time_split"):
        assert k in d, k
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["unit"] == "moves/s" and d["dtype"].startswith("f32")
    assert "workload" in d["config"] and "model" not in d["config"] and "HIP graph" in d["config"]["loop"]
    assert d["config"]["ticks_timed_rank0"] > d["config"]["ticks_event_timed_rank0"] > 0
    r_ = d["roofline"]
    assert r_["bound"] == "mfma" and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-9 and r_["ms_per_launch"] > 0
    assert abs(r_["achieved"] - r_["flop_per_launch"] / (r_["ms_per_launch"] * 1e-3) / 1e12) < 1e-6 * r_["achieved"]
    plies = d["ms_per_step"] * 1e-3 * d["steps"] * (d["value"] or d["opening_phase_moves_per_s"])
    assert abs(plies - 2 * 512) < 0.25 * 2 * 512                 # a step commits about one ply per game
    assert "rank0_handoff_ms_per_step" in d["config"] and d["config"]["rank0_handoff_ms_per_step"] < 5.0
    assert abs(d["time_split"]["outside_kernels_us_per_tick"]) < 100


def test_plain_python_gpus_n_spawns_ranks_and_a_failing_rank_ends_the_job():
    """`python bench.py --gpus 2` with no launcher around it must start its own ranks (it used to exit with "launch
    with torch.distributed.run").  Without a GPU every spawned rank stops at "needs a HIP device": the launcher must then end with a
    non-zero code — not hang on a half-formed group — and both ranks must have been started."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-box form of the spawn test; tests/test_gpu_multirank.py runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode()
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in err
    assert err.count("needs a HIP device") >= 1                  # at least the first rank to fail got that far; the rest were stopped
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


def test_power_sampler_without_a_sensor_reports_nothing():
    """bench.PowerSampler reads the amdgpu hwmon node of the device's PCI address; a box without one (this container) must give
    roofline.power = None, never an exception inside the timed region."""
    import types
    import bench
    ps = bench.PowerSampler(types.SimpleNamespace(pci_domain_id=0xffff, pci_bus_id=0xfe, pci_device_id=0x1f))
    assert ps.paths is None
    ps.start()
    assert ps.stop() is None
    assert bench.PowerSampler(object()).stop() is None              # a properties object without the PCI fields

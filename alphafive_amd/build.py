"""In-tree build of the HIP engine for gfx950 (hipcc cross-compiles without a GPU).

    python -m alphafive_amd.build            # builds alphafive_amd/_lib/*.so
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
LIBDIR = os.path.join(PKG, "_lib")
ARCH = "gfx950"

HIPCC_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(REPO, "include"),
               "-I", os.path.join(PKG, "csrc")]

# name -> (sources, extra flags).  -ffp-contract=off: the tree engine's arithmetic must be
# exactly as written; the net kernels are free to contract (fp32 MFMA is an fma chain).
TARGETS = {
    "libaf_hip.so": (["csrc/af_engine.hip"], ["-ffp-contract=off"]),
    "libaf_net.so": (["csrc/af_net.hip", "csrc/af_conv_f16s.hip"], []),
    "libaf_tower.so": (["csrc/af_tower_bf16.hip"], []),
    "libaf_replay.so": (["csrc/af_replay.hip"], []),
}


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    deps = list(srcs) + [os.path.join(REPO, "include", f) for f in os.listdir(os.path.join(REPO, "include"))]
    deps += [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith(".h")]
    return any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build_all(force=False, verbose=True):
    """Build every stale library; the (independent) hipcc invocations run side by side."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    built, jobs = [], []
    for name, (srcs, extra) in TARGETS.items():
        out = os.path.join(LIBDIR, name)
        srcs = [os.path.join(PKG, s) for s in srcs]
        if force or _stale(out, srcs):
            cmd = [hipcc] + HIPCC_FLAGS + extra + ["-o", out] + srcs
            if verbose:
                print("[alphafive_amd.build]", " ".join(cmd), flush=True)
            jobs.append(cmd)
        built.append(out)
    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for rc, cmd in zip(ex.map(subprocess.call, jobs), jobs):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, cmd)
    return built


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)

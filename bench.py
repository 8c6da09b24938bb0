#!/usr/bin/env python3
"""bench.py — headline benchmark: particle-element-steps/s, 100-element FODO linac, 1e6 particles / GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1` (one rank per GPU, RCCL); under a launcher (RANK / WORLD_SIZE set, the driver's form) it asserts
that the world size equals N.

Workload (BASELINE.json configs[1], "C2"): 25 x [Quad(0.2, +4.2), Drift(0.8), Quad(0.2, -4.2), Drift(0.8)],
ParticleBeam.from_parameters defaults, fp32, 1e6 particles PER RANK (weak scaling: the particle axis shards with no
data-path collective; only the global beam moments cross ranks — one all-gather of 29 doubles per rank).

One timed "step" = `Segment.track_elementwise(beam)` — all 100 elements applied one after the other, one pass over HBM per
element, NO map merging — followed by the global beam moments of the outgoing beam. `value` = N_gpus * 1e6 * 100 / t_step.
The reference's own semantics (`Segment.track`: merge the 100 maps, one pass) and the fused in-register variant are timed as
well ("modes") — N*E/t is not a bandwidth measure for those.

`roofline`: the dominant kernel is the linear apply (apply_tile_kernel<float,2,0> at 1e6 particles, apply_wave_kernel beyond;
100 launches per step). The line LEADS with the true-HBM figure: the same kernel at 1.6e7 particles (two 448 MB buffers that
cannot live in the 256 MiB Infinity Cache), average launch duration from HIP events on the launch stream, `achieved` =
56 B x particles / duration, `frac` against the 8 TB/s spec and `frac_vs_copy_ceiling` against the 6.29 TB/s a float4 copy
reaches on this part. `workload_size` holds the same measurement at the benchmark's 1e6 particles (events around the 100-launch
run of EVERY timed step, next to the rocprofv3 kernel-trace average committed under profiles/): its 56 MB working set is
Infinity-Cache resident, so it is labelled "infinity-cache / hbm" and is not an HBM rate.

`configs` (1 GPU only): the other BASELINE.json configs at full size — C1 merged track, C3 k1 scan, C4 space-charge linac,
C5 backward — each with its own algorithmic-byte accounting. `cpu_baseline` (rank 0, N = 1): the C oracle (OpenMP, pinned
to the physical cores) and a plain PyTorch-CPU `(N,7) @ (7,7)` loop (the reference's element.py:182) on the same workload,
timed in a subprocess with its own thread settings.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import csv
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PARTICLES = 1_000_000
N_CELLS = 25
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
APPLY_KERNEL = "apply_tile_kernel<float, 2, 0>"
PROFILE_CSV = os.path.join(ROOT, "profiles", "r06_kernel_stats.csv")
COPY_CEILING_GBS = 6290.0   # MI355X_MICROARCH.md: measured float4 copy (79 % of the 8 TB/s spec)
#: sigma_x behind the 100 elements of the beam of rank 0 (torch.manual_seed(1234), from_parameters defaults, 1e6 particles, fp32):
#: the same digits in every driver run since round 1 (BENCH_r01..r03). bench.py refuses to print a line when its step no longer
#: produces it.
EXPECTED_SIGMA_X_RANK0 = 1.9558527096402663e-4


def build_fodo(ca, torch, device, dtype):
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
    kw = {"dtype": dtype, "device": device}
    els = []
    for _ in range(N_CELLS):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2), **kw), ca.Drift(t(0.8), **kw),
                ca.Quadrupole(t(0.2), k1=t(-4.2), **kw), ca.Drift(t(0.8), **kw)]
    return ca.Segment(els)


def timed(torch, dist, fn, steps, warmup, world):
    multi = world > 1 or (dist.is_available() and dist.is_initialized())   # a one-rank group (forced collectives) as well
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


# ---------------------------------------------------------------------------------------------------------------- CPU side
def usable_cores() -> int:
    """Cores this process may actually keep busy: physical cores, capped by the affinity mask and by the container's CPU quota
    (cgroup cpu.max / cfs_quota). The GPU boxes of this pool show 256 logical CPUs behind a 16-CPU quota: a thread per core there
    is throttled to a crawl (measured with benchmarks/cpu_probe.py: the C port drops from 1.9e9 steps/s at 16 threads to 6.5e8 at
    128 and 8e6 at 256)."""
    try:
        import psutil

        n = psutil.cpu_count(logical=False) or 0
    except Exception:
        n = 0
    if not n:
        n = max(1, (os.cpu_count() or 2) // 2)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cpu_baseline_worker(spec: str) -> dict:
    """One leg of the CPU baseline, in a process of its own whose OMP_NUM_THREADS is the leg's thread count.
    spec = "<port|torch>:<elementwise|merged>:<n_elements>:<seconds>"."""
    import numpy as np

    which, mode, n_elements, seconds = spec.split(":")
    n_elements, seconds = int(n_elements), float(seconds)
    threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    from oracle import chx_oracle as oracle

    E = 1e8
    f = np.float32
    cell = [oracle.build_rmatrix("quadrupole", [f(0.2), f(4.2), 0, 0, 0], E), oracle.build_rmatrix("drift", [f(0.8)], E),
            oracle.build_rmatrix("quadrupole", [f(0.2), f(-4.2), 0, 0, 0], E), oracle.build_rmatrix("drift", [f(0.8)], E)]
    maps64 = (cell * N_CELLS)[:n_elements]
    maps = np.concatenate([m.astype(np.float32) for m in maps64])
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((N_PARTICLES, 7)) * [175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3, 0]).astype(np.float32)
    x[..., 6] = 1.0
    reps = 0
    if which == "port":
        out, tmp, xs = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        # first touch of every buffer happens inside the OpenMP loops (NUMA-local pages on a multi-socket host)
        oracle.track_elementwise(x, np.eye(7, dtype=np.float32)[None], xs, tmp)
        if mode == "elementwise":
            step = lambda: oracle.track_elementwise(xs, maps, out, tmp)  # noqa: E731
        else:   # the reference's own semantics (segment.py:534-547): compose the E maps, ONE pass over the particles
            step = lambda: oracle.track_elementwise(xs, oracle.compose(maps64).astype(np.float32), out, tmp)  # noqa: E731
        step()
        t0 = time.perf_counter()
        while True:
            step()
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                break
    else:
        import torch

        torch.set_num_threads(threads)
        xt = torch.from_numpy(x)
        mt = [torch.from_numpy(np.ascontiguousarray(m)) for m in maps]
        with torch.no_grad():
            def elementwise():          # element.py:182 `particles @ tm.mT`, one matmul (and one fresh output) per element
                y = xt
                for m in mt:
                    y = y @ m.mT
                return y

            def merged():               # segment.py:534-547: the 7x7 product first, one matmul over the particles
                tm = mt[0]
                for m in mt[1:]:
                    tm = m @ tm
                return xt @ tm.mT

            step = elementwise if mode == "elementwise" else merged
            step()
            step()
            t0 = time.perf_counter()
            while True:
                step()
                reps += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    break
    return {"value": N_PARTICLES * n_elements * reps / el, "threads": threads, "reps": reps, "seconds": el}


def cpu_baseline(n_elements: int) -> dict:
    """The CPU figures of the line (rank 0, N = 1), each leg in its own subprocess: the C oracle (OpenMP) and a plain PyTorch-CPU
    restatement of the reference's tensor code, both element by element (the headline's semantics) and merged (the reference's
    `Segment.track`: BASELINE.md section 2 reports both modes). Thread counts are swept up to the cores this process may use
    (`usable_cores`: the container's CPU quota, not the host's core count); the best of each leg is reported. ~25 s in total."""
    cores = usable_cores()

    def leg(which, mode, threads, seconds, bind):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="",
                   CUDA_VISIBLE_DEVICES="")
        env.pop("OMP_PROC_BIND", None)
        env.pop("OMP_PLACES", None)
        if bind:
            env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", f"{which}:{mode}:{n_elements}:{seconds}"],
                             env=env, capture_output=True, text=True, timeout=180)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not line:
            raise RuntimeError(out.stderr[-300:])
        return json.loads(line[-1])

    def best(which, mode, counts, seconds, bind=False):
        runs = {}
        for threads in counts:
            try:
                runs[threads] = leg(which, mode, threads, seconds, bind)
            except Exception as exc:  # noqa: BLE001
                runs[threads] = {"value": 0.0, "error": f"{type(exc).__name__}: {exc}"[:200]}
        top = max(runs, key=lambda k: runs[k]["value"])
        return top, runs

    counts = sorted({max(1, cores // 2), cores, 2 * cores})
    try:
        t_port, port = best("port", "elementwise", counts, 4.0)
        res = {"value": port[t_port]["value"], "unit": "particle-element-steps/s", "cores": t_port, "kind": "port",
               "usable_cores": cores, "host_logical_cpus": os.cpu_count(),
               "by_threads": {str(k): v["value"] for k, v in port.items()},
               "sample": f"{port[t_port].get('reps')} x (1e6 particles x {n_elements} elements, element by element, fp32 fma chain, C "
                         f"oracle with OpenMP, {t_port} threads — best of {counts}; the container's CPU quota is {cores} of the "
                         f"host's {os.cpu_count()} logical CPUs), {port[t_port].get('seconds', 0):.1f} s"}
        _, merged = best("port", "merged", [t_port], 2.0)
        res["merged"] = {"value": merged[t_port]["value"], "unit": "particle-element-steps/s", "threads": t_port,
                         "sample": "the reference's Segment.track semantics: the 100 maps composed, ONE pass over the 1e6 particles "
                                   "(N*E/t is not a bandwidth measure in this mode)"}
        tcounts = sorted({max(1, cores // 2), cores})
        t_t, tor = best("torch", "elementwise", tcounts, 3.0)
        _, tmerged = best("torch", "merged", [t_t], 2.0)
        import torch

        res["torch_cpu_matmul"] = {"value": tor[t_t]["value"], "unit": "particle-element-steps/s", "threads": t_t,
                                   "torch": torch.__version__, "by_threads": {str(k): v["value"] for k, v in tor.items()},
                                   "merged": tmerged[t_t]["value"],
                                   "sample": f"(1e6 x 7 fp32) @ (7 x 7).mT per element over {n_elements} elements (element.py:182) and "
                                             f"merged (segment.py:534-547), best of thread counts {tcounts}; torch's MKL sgemm on "
                                             "this host — BASELINE.md section 2 has the reference itself on an 8-core Intel "
                                             "container: 1.3e9 element by element, 3.1e10 merged"}
        return res
    except Exception as exc:  # the GPU numbers stand on their own; say why the baseline is missing
        return {"value": None, "unit": "particle-element-steps/s", "cores": cores, "kind": "port",
                "sample": f"cpu baseline failed: {type(exc).__name__}: {exc}"}


# ---------------------------------------------------------------------------------------------------------------- launch
def self_launch(args) -> None:
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and not args.one_device_gloo:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node has {have}; "
                         f"run with --gpus {max(have, 1)} or on a larger node\n")
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.execvpe(sys.executable, cmd, env)


def rocprof_average_ms() -> float | None:
    """Average duration of the apply kernel in the tracked rocprofv3 kernel-trace summary of this same command."""
    try:
        for row in csv.DictReader(open(PROFILE_CSV)):
            if APPLY_KERNEL.replace(" ", "") in row["Name"].replace(" ", ""):
                return float(row["AverageNs"]) * 1e-6
    except Exception:
        pass
    return None


def live_pmc_traffic() -> dict | None:
    """HBM bytes per launch of the apply kernel measured IN THIS RUN: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE: separate
    `--pmc` passes with the kernel trace only, as MI355X_MICROARCH.md prescribes) over profiles/traffic_probe.py in subprocesses,
    counters in KiB, FETCH_SIZE x 2 for wide coalesced reads on gfx950. None when rocprofv3 is missing, fails or is switched off
    (CHX_BENCH_NO_PMC=1); the tracked figure of profiles/apply_traffic.json stands then."""
    import shutil
    import tempfile

    if os.environ.get("CHX_BENCH_NO_PMC", "0") == "1" or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None     # (switched off, or this process itself runs under the profiler: no nested sessions)
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    out = {}
    try:
        with tempfile.TemporaryDirectory(prefix="chx_pmc_", dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                proc = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "probe", "--",
                                       sys.executable, os.path.join(ROOT, "profiles", "traffic_probe.py")], capture_output=True, text=True,
                                      timeout=240, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"})
                if proc.returncode != 0:
                    return None
                found = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
                if not found:
                    return None
                per = {}
                for row in csv.DictReader(open(found[0])):
                    name = row["Kernel_Name"]
                    if row["Counter_Name"] != counter or not ("apply_tile_kernel" in name or "apply_wave_kernel" in name):
                        continue
                    ppt = int(name.split("<")[1].split(",")[1])
                    grid = int(row["Grid_Size"])
                    n_part = 1_000_000 if (grid, ppt) == (500224, 2) else grid * ppt
                    per.setdefault(n_part, []).append(float(row["Counter_Value"]))
                for n_part, vals in per.items():
                    out.setdefault(n_part, {})[counter] = sum(vals) / len(vals) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
    except Exception:   # noqa: BLE001  (a diagnostic: the timing stands without it)
        return None
    small, big = out.get(1_000_000, {}), out.get(16_000_000, {})
    if "FETCH_SIZE" not in small or "WRITE_SIZE" not in small:
        return None
    res = {"hbm_bytes_per_launch": small["FETCH_SIZE"] + small["WRITE_SIZE"], "fetch_bytes": small["FETCH_SIZE"], "write_bytes": small["WRITE_SIZE"]}
    if "FETCH_SIZE" in big and "WRITE_SIZE" in big:
        res["streaming_hbm_bytes_per_launch"] = big["FETCH_SIZE"] + big["WRITE_SIZE"]
    return res


def event_timed_elementwise(torch, seg, beam, steps, warmup):
    """Average milliseconds of one `chx_track_elementwise` run (E launches) from HIP events on the launch stream."""
    for _ in range(warmup):
        seg.track_elementwise(beam, fused=False)
    pairs = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        seg.track_elementwise(beam, fused=False)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)


# ---------------------------------------------------------------------------------------------------------------- configs
def other_configs(ca, torch, device, only=None) -> dict:
    """C1 / C3 / C4 / C5 of BASELINE.json at full size on one GPU (seeded synthetic inputs), ms per call. `only`: names to run."""
    from benchmarks import run_configs as rc

    rc.DEV = device
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # a failing side config must not take the headline down with it
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    def c1():
        r = rc.c1()
        return {"workload": "C1: ARES 13-element segment (README), 1e4 particles, fp64, Segment.track (maps merged)",
                "ms_per_track": r["track_ms"], "ms_track_plus_cic_reading": r["track_plus_cic_reading_ms"],
                "roofline": {"bound": "host launch", "algorithmic_bytes": 1e4 * 112.0,
                             "note": "1.1 MB per track: floor 0.14 us of HBM time, the call is launch / host bound"}}

    def c3():
        r = rc.c3()
        nbytes = r["output_GB"] * 1e9 + 2.8e6
        return {"workload": "C3: k1 scan B=4096 x N=1e5 on the ARES EA subcell, fp32, shared beam, ONE GPU",
                "ms_per_track": r["track_ms"], "ms_all_moments": r["all_moments_ms"],
                "ms_track_moments_fused": r["fused_track_moments_ms"],
                "ms_track_moments_algebraic": r["algebraic_track_moments_ms"],
                "particle_element_steps_per_s": r["steps_per_s"],
                "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / (r["track_ms"] * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (r["track_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "28 B written per (setting, particle) + the 2.8 MB beam read once (SURVEY 8d)"}}

    def c4_particle_kernels():
        # average durations of the chain's per-particle kernels in the tracked rocprofv3 summary (run under the profiler with the
        # Green chain on the side stream: durations include what the overlap costs them)
        import csv
        path = os.path.join(ROOT, "profiles", "r06_c4_kernel_stats.csv")
        avg = {}
        try:
            for row in csv.DictReader(open(path)):
                for key, name in (("sc_tile_deposit_kernel", "sc_tile_deposit_kernel"), ("sc_tile_schedule_kernel", "sc_tile_schedule_kernel"),
                                  ("sc_tile_particle_kernel", "sc_tile_particle32_kernel")):   # (round 6: the float32 beam's gather pass)
                    if name in row["Name"]:
                        avg[key] = float(row["AverageNs"]) * 1e-3
        except OSError:
            return None
        if "sc_tile_deposit_kernel" not in avg or "sc_tile_particle_kernel" not in avg:
            return None
        # (round 5: the deposit's bookkeeping rides in the first FFT pass; a schedule kernel only runs for the kicks without riders)
        dep = avg["sc_tile_deposit_kernel"]
        gat = avg["sc_tile_particle_kernel"]
        rate = 84.0 * N_PARTICLES / ((dep + gat) * 1e-6) / 1e9
        return {"bytes_per_particle": 84.0, "deposit_us_profile": dep, "gather_us_profile": gat, "achieved_GBs": rate,
                "frac": rate / HBM_PEAK_GBS, "source": "profiles/r06_c4_kernel_stats.csv",
                "launches_per_kick": 13, "note_launches": "deposit, five charge-FFT passes, gather on the main stream; corner table, far "
                "field, three Green FFT passes and the next run's map on the side stream (profiles/r05_c4_timeline.txt); round 6: the "
                "gather pass evaluates a float32 beam's particle step in float32 (sc_tile_particle32_kernel, profiles/r06_c4_gather.md)"}

    def c4_fp32_kick_error():
        # the float32 kick of C4's first SpaceChargeKick against the reference's float64 run of the same particles (committed sample,
        # tests/golden/fullsize_c4.npz): max / rms error per momentum coordinate as a fraction of the kick amplitude — a drift of
        # the kernels' arithmetic shows up here (tests/test_gpu_fullsize.py bounds it at 5e-4)
        try:
            import numpy as np

            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import fullsize_inputs as fi

            g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_c4.npz"))
            dt = torch.float32
            kw = {"dtype": dt, "device": device}
            tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
            beam = ca.ParticleBeam(torch.from_numpy(fi.c4_particles()).to(dt).to(device), tt(fi.C4_ENERGY),
                                   particle_charges=torch.from_numpy(fi.c4_charges()).to(dt).to(device), species=ca.Species("electron", **kw))
            b1 = ca.Drift(tt(0.1), **kw).track(beam)
            b2 = ca.SpaceChargeKick(tt(0.2), grid_shape=fi.C4_GRID, **kw).track(b1)
            sl = slice(None, None, fi.C4_SAMPLE_STRIDE)
            got = (b2.particles[sl].double() - b1.particles[sl].double()).cpu().numpy()
            ref = g["kick1_out_sample"] - g["kick1_in_sample"]
            kick = np.max(np.abs(ref), axis=0)
            err = np.max(np.abs(got - ref), axis=0) / kick
            rms = np.sqrt(np.mean((got - ref) ** 2, axis=0)) / kick
            return {"max_over_kick_amplitude": {"px": float(err[1]), "py": float(err[3]), "delta": float(err[5])},
                    "rms_over_kick_amplitude": {"px": float(rms[1]), "py": float(rms[3]), "delta": float(rms[5])},
                    "against": "the reference's float64 kick of the same 1e6 particles (tests/golden/fullsize_c4.npz, every "
                               f"{fi.C4_SAMPLE_STRIDE}th particle)", "test_bound": 5e-4}
        except Exception as exc:  # noqa: BLE001
            return {"error": f"{type(exc).__name__}: {exc}"[:300]}

    def c4():
        r = rc.c4()
        # SURVEY 8d per kick: 84 B per particle + ~1.6 GB of grid / dense-FFT traffic (the pruned solver moves ~0.35 GB)
        per_kick = 84.0 * N_PARTICLES + 1.6e9
        in_track = (r["track_ms"] - 0.0) / 10.0          # a kick of the chain (tile-ordered beam), the linear runs folded in
        return {"workload": "C4: 50-element linac, 10 SpaceChargeKicks on 128^3, 1e6 particles, fp32",
                "ms_per_track": r["track_ms"], "ms_per_kick_in_track": in_track, "ms_per_isolated_kick": r["single_kick_ms"],
                "particle_element_steps_per_s": r["steps_per_s"], "fp32_kick_error": c4_fp32_kick_error(),
                "roofline": {"bound": "hbm (model floor)", "algorithmic_bytes_per_kick": per_kick,
                             "ratio_to_model_floor": per_kick / (in_track * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "particle_kernels": c4_particle_kernels(),
                             "note": "the byte count is the SURVEY 8d model (three dense 256^3 transforms); the pruned, "
                                     "symmetry-aware solver moves about a fifth of it, so `ratio_to_model_floor` is a ratio to "
                                     "the model's floor time, not an achieved bandwidth. `particle_kernels` is the achieved rate "
                                     "of the per-particle half (tile deposit + gather on the tile-ordered beam) from the "
                                     "tracked rocprofv3 summary."}}

    def c5():
        r = rc.c5()
        # forward: apply 56 B + the Screen's record of the beam 36 + 28 B (same pass) + moments 32 B per particle ~ 144 B; backward: 7x7 algebra on the
        # incoming beam's (memoised) moments — no particle pass (the particle-sized backward moved 144 B more). 144 B is the model every
        # round's fraction was quoted on; since round 6 the record's moments come out of the particle pass itself, so the step MOVES
        # 100 B per particle (rows 28 + charges, weights 8 in; rows 28, record 28 + 8 out): `moved_bytes` / `frac_of_moved_bytes`
        nbytes = 144.0 * N_PARTICLES
        moved = 100.0 * N_PARTICLES
        res = {"workload": "C5: d sigma_x(screen)/d k1, [Drift, Quad(k1), Drift, Screen], 1e6 particles, fp32, fwd+bwd",
               "ms_fwd_bwd": r["fwd_bwd_ms"], "ms_fwd_bwd_first_50_steps": r.get("fwd_bwd_ms_first_50_steps"),
               "steps_timed": r.get("steps_timed"), "sigma_x": r["sigma_x"], "dsigma_x_dk1": r["dk1"],
               "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / (r["fwd_bwd_ms"] * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (r["fwd_bwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "moved_bytes": moved, "frac_of_moved_bytes": moved / (r["fwd_bwd_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "eager step: four launches (preparation, particle pass + the record's moment sums, finalize; backward: "
                                    "the builders' VJP forming dL/dC itself) and ONE autograd node from sigma_x to k1, in C++ "
                                    "(cheetah_amd._chxtorch RunMomentEntry); bound by the host: ~80 us forward + the autograd engine "
                                    "(60 us for a one-node graph on these hosts, benchmarks/c5_variants.py)"}}
        # The same step captured once into a device graph (torch.cuda.CUDAGraph = hipGraph) and replayed: what an optimisation loop
        # that keeps its tensors in place can run. In a process of its own: a capture needs Parameters that have never seen a
        # backward pass on the default stream, and a failed capture must not take this line down.
        import subprocess

        try:
            proc = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "c5_graph.py")], capture_output=True, text=True,
                                  timeout=300, cwd=ROOT)
            line = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"c5_graph"')]
            if proc.returncode != 0 or not line:
                raise RuntimeError(f"rc {proc.returncode}: {proc.stderr[-300:]}")
            g = json.loads(line[-1])["c5_graph"]
            ms = g["graph_replay_us"] * 1e-3
            res["graph_replay"] = {"ms_fwd_bwd": ms, "ms_fwd_bwd_eager_same_process": g["eager_us"] * 1e-3,
                                   "loss_equals_eager": abs(g["loss"] - g["loss_eager"]) <= 1e-6 * abs(g["loss_eager"]),
                                   "grad_equals_eager": abs(g["grad"] - g["grad_eager"]) <= 1e-5 * abs(g["grad_eager"]),
                                   "roofline": {"bound": "hbm", "algorithmic_bytes": nbytes, "achieved": nbytes / (ms * 1e-3) / 1e9,
                                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "moved_bytes": moved, "frac_of_moved_bytes": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                   "note": "benchmarks/c5_graph.py: forward + backward captured after a warm-up on a side stream; k1 "
                                           "is updated in place between replays and the replayed kernels read it through its pointer"}
        except Exception as exc:  # noqa: BLE001
            res["graph_replay"] = {"error": str(exc)[:300]}
        return res

    def dkd():
        # 100-element FODO tracked element by element with the Bmad-X drift-kick-drift maps
        import time as _t

        dt = torch.float32
        kw = {"dtype": dt, "device": device}
        tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
        els = []
        for _ in range(N_CELLS):
            els += [ca.Quadrupole(tt(0.2), k1=tt(4.2), tracking_method="drift_kick_drift", **kw),
                    ca.Drift(tt(0.8), tracking_method="drift_kick_drift", **kw),
                    ca.Quadrupole(tt(0.2), k1=tt(-4.2), tracking_method="drift_kick_drift", **kw),
                    ca.Drift(tt(0.8), tracking_method="drift_kick_drift", **kw)]
        beam = ca.ParticleBeam.from_parameters(num_particles=N_PARTICLES, **kw)
        res = {"workload": "100-element FODO, drift_kick_drift (Bmad-X) tracking of every element, 1e6 particles, fp32: "
                           "Segment.track (no element of this lattice can be merged with another; one chx_dkd_chain call carries "
                           "every particle through the 100 maps in registers: one read and one write of the beam) and the "
                           "elements' own track() one after the other (100 passes over HBM)"}
        seg = ca.Segment(els)

        def by_segment():
            return seg.track(beam)

        def one_by_one():
            b = beam
            for e in els:
                b = e.track(b)
            return b

        for label, prec in (("mixed_arithmetic_default", "mixed"), ("float64_arithmetic", "double"), ("float32_arithmetic", "storage")):
            for e in els:
                e.dkd_precision = prec
            res[label] = {}
            for how, run in (("segment_track", by_segment), ("element_by_element", one_by_one)):
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                t0 = _t.perf_counter()
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                ms = (_t.perf_counter() - t0) / 5 * 1e3
                res[label][how] = {"ms_per_track": ms, "particle_element_steps_per_s": N_PARTICLES * len(els) / (ms * 1e-3)}
                if how == "element_by_element":          # 56 B per particle and element through HBM
                    res[label][how]["achieved_GBs"] = 56.0 * N_PARTICLES * len(els) / (ms * 1e-3) / 1e9
                else:                                    # 56 B per particle and TRACK: bound by the maps' arithmetic
                    res[label][how]["hbm_bytes_per_track"] = 56.0 * N_PARTICLES
                    res[label][how]["bound"] = "valu (particles in registers across the run)"
        res["note"] = ("dkd_precision: mixed (default of float32 beams: tau / delta in fp64, the rest in float32), double, storage; "
                       "per-element kernel times and measured errors against the reference's float64 run: "
                       "profiles/r04_dkd_precision.md, tests/test_gpu_bench_parity.py")
        return res

    def second_order():
        # the same FODO with every element's second-order map (element.py:195-228): consecutive elements in one chx_second_order_chain
        import time as _t

        dt = torch.float32
        kw = {"dtype": dt, "device": device}
        tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
        so = {"tracking_method": "second_order"}
        els = []
        for _ in range(N_CELLS):
            els += [ca.Quadrupole(tt(0.2), k1=tt(4.2), **so, **kw), ca.Drift(tt(0.8), **so, **kw),
                    ca.Quadrupole(tt(0.2), k1=tt(-4.2), **so, **kw), ca.Drift(tt(0.8), **so, **kw)]
        seg = ca.Segment(els)
        beam = ca.ParticleBeam.from_parameters(num_particles=N_PARTICLES, **kw)

        def one_by_one():
            b = beam
            for e in els:
                b = e.track(b)
            return b

        times = {}
        with torch.no_grad():
            for how, run in (("segment_track", lambda: seg.track(beam)), ("element_by_element", one_by_one)):
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                t0 = _t.perf_counter()
                for _ in range(10):
                    run()
                torch.cuda.synchronize()
                times[how] = (_t.perf_counter() - t0) / 10 * 1e3
        ms = times["segment_track"]
        return {"workload": "100-element FODO, second_order tracking of every element, 1e6 particles, fp32: Segment.track (one "
                            "chx_second_order_chain call: every particle through the 100 maps in registers, one read and one write "
                            "of the beam) and the elements' own track() one after the other (100 passes over HBM)",
                "ms_per_track": ms, "particle_element_steps_per_s": N_PARTICLES * len(els) / (ms * 1e-3),
                "hbm_bytes_per_track": 56.0 * N_PARTICLES, "bound": "valu (particles in registers across the run)",
                "element_by_element": {"ms_per_track": times["element_by_element"],
                                       "achieved_GBs": 56.0 * N_PARTICLES * len(els) / (times["element_by_element"] * 1e-3) / 1e9}}

    def diagnostics():
        # lattices with things between the magnets that read or thin the beam: active BPMs, apertures, cavities — one stretch call
        # each (chx_lattice_track_diag / chx_parameter_lattice_track), ParticleBeam of 1e5 particles and ParameterBeam. Lattices and
        # beams are DATA drawn on the host from fixed seeds (benchmarks/diagnostics_inputs.py); every entry is checked against the
        # REFERENCE's float64 run of the same data (tests/golden/bench_diagnostics.json) before its time is reported.
        import time as _t

        from benchmarks import diagnostics_inputs as di

        dt = torch.float32
        kw = {"dtype": dt, "device": device}
        tt = lambda v: torch.tensor(v, **kw)  # noqa: E731
        with open(os.path.join(ROOT, "tests", "golden", "bench_diagnostics.json")) as fh:
            expected = json.load(fh)
        checks = {"entries": 0, "values": 0, "worst_relative_error": 0.0}

        def timed_us(fn, reps=30):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (_t.perf_counter() - t0) / reps * 1e6

        def check(entry, variant, seg, out, rows=None):
            """sigma_x, sigma_y, energy, the surviving weight and the last monitor's reading of what was just tracked against the
            reference's numbers: float32 tracking of ~100 elements against float64 — 2e-4 of a beam size, 1e-6 of the energy."""
            want = expected[entry][variant]
            bpms = [e for e in seg.elements if isinstance(e, ca.BPM)]
            got = {}
            for tag, b in ([("", None)] if rows is None else [(f"_row{r}", r) for r in rows]):
                pick = lambda t: float(t if (b is None or t.dim() == 0) else t[b])  # noqa: E731
                got["sigma_x" + tag], got["sigma_y" + tag], got["energy" + tag] = pick(out.sigma_x), pick(out.sigma_y), pick(out.energy)
                if bpms:
                    r = bpms[-1].reading
                    got["last_reading_x" + tag] = float(r[..., 0] if b is None or r.dim() == 1 else r[b, 0])
            if isinstance(out, ca.ParticleBeam):
                w = out.survival_probabilities
                got["survived"] = float(w.sum() if w.dim() == 1 else w[0].sum())
            assert set(got) == set(want), (entry, variant, sorted(set(got) ^ set(want)))
            for key, v in want.items():
                if key.startswith("last_reading"):
                    scale, tol = max(abs(v), want["sigma_x" + key[len("last_reading_x"):]]), 2e-4      # (a mean: against the beam size)
                elif key.startswith("energy") or key == "survived":
                    scale, tol = abs(v), 1e-6
                else:
                    scale, tol = abs(v), 2e-4
                err = abs(got[key] - v) / scale
                checks["worst_relative_error"] = max(checks["worst_relative_error"], err)
                checks["values"] += 1
                if not err < tol:
                    raise AssertionError(f"DIAGNOSTICS_LATTICES {entry}/{variant}: {key} = {got[key]!r}, the reference has {v!r}")
            checks["entries"] += 1

        x = di.particles().to(device)
        E0 = tt(1e8)
        beam = ca.ParticleBeam(x, E0, **kw)
        small = ca.ParticleBeam(x[:di.N_SMALL].contiguous(), E0, **kw)
        mu, cov = di.parameter_beam_moments()
        pbeam = ca.ParameterBeam(mu.to(**kw), cov.to(**kw), E0, **kw)
        many = ca.ParticleBeam(x[:di.N_SMALL].unsqueeze(0).repeat(16, 1, 1).contiguous(), E0, **kw)
        res = {"workload": "100-element lattices with 25 active BPMs / 25 active apertures and a 16-cell cavity linac, fp32: us per "
                           "Segment.track (ParticleBeam of 1e5 particles; ParameterBeam; 16 beams of 1e4 particles in one ParticleBeam); an orbit "
                           "response of 25 cells, each corrector angle a (64,) tensor (ParameterBeam; one 1e4-particle beam shared by the rows); the "
                           "cavity linac at 64 beam energies, with the phase of every cavity a (64,) tensor, and with two cavities switched off. "
                           "Every entry's outgoing beam (sigma_x, sigma_y, energy, surviving weight, last monitor reading) is compared with the "
                           "REFERENCE's float64 run of the same seeded data (tests/golden/bench_diagnostics.json) before its time is reported. "
                           "scan_4096_settings_x_1e5_shared_particles: C3's size through six cells with one / six active monitors, ms per "
                           "Segment.track and GB/s of the (4096, 1e5, 7) result, rows 0 and 4095 checked against their own single-row track"}
        with torch.no_grad():
            for name, specs in (("bpm_lattice", di.bpm_lattice()), ("aperture_lattice", di.aperture_lattice()), ("cavity_linac", di.cavity_linac())):
                seg = di.segment(ca, specs, kw)
                check(name, "particle_beam", seg, seg.track(beam))
                res[name] = {"particle_beam_us": timed_us(lambda: seg.track(beam)),
                             "sixteen_beams_us": timed_us(lambda: seg.track(many))}
                if name != "aperture_lattice":      # (an aperture only warns for a ParameterBeam)
                    check(name, "parameter_beam", seg, seg.track(pbeam))
                    res[name]["parameter_beam_us"] = timed_us(lambda: seg.track(pbeam))
            # an orbit response: every corrector's angle a (64,) tensor, 25 monitors; all 64 settings in one stretch call, for a
            # ParameterBeam and for ONE ParticleBeam of 1e4 particles shared by the settings
            seg = di.segment(ca, di.orbit_response(), kw)
            check("orbit_response_64_settings", "parameter_beam", seg, seg.track(pbeam), rows=(0, 63))
            check("orbit_response_64_settings", "particle_beam_1e4", seg, seg.track(small), rows=(0, 63))
            res["orbit_response_64_settings"] = {"parameter_beam_us": timed_us(lambda: seg.track(pbeam)),
                                                 "particle_beam_1e4_us": timed_us(lambda: seg.track(small))}
            # the 16-cell cavity linac under scans and with cavities switched off: 64 beam energies; the phase of every cavity a
            # (64,) tensor; two cavities at voltage 0 (skippable, drift-like) — each one stretch call
            energies = tt(di.energies())
            e_pb = ca.ParameterBeam(mu.to(**kw), cov.to(**kw), energies, **kw)
            e_beam = ca.ParticleBeam(small.particles, energies, **kw)
            seg = di.segment(ca, di.cavity_linac(), kw)
            check("linac_64_energies", "parameter_beam", seg, seg.track(e_pb), rows=(0, 63))
            check("linac_64_energies", "particle_beam_1e4", seg, seg.track(e_beam), rows=(0, 63))
            res["linac_64_energies"] = {"parameter_beam_us": timed_us(lambda: seg.track(e_pb)),
                                        "particle_beam_1e4_us": timed_us(lambda: seg.track(e_beam))}
            seg = di.segment(ca, di.cavity_linac(phase=di.phases()), kw)
            check("linac_64_phases_of_every_cavity", "parameter_beam", seg, seg.track(pbeam), rows=(0, 63))
            check("linac_64_phases_of_every_cavity", "particle_beam_1e4", seg, seg.track(small), rows=(0, 63))
            res["linac_64_phases_of_every_cavity"] = {"parameter_beam_us": timed_us(lambda: seg.track(pbeam)),
                                                      "particle_beam_1e4_us": timed_us(lambda: seg.track(small))}
            seg = di.segment(ca, di.cavity_linac(off=(5, 9)), kw)
            check("linac_two_cavities_off", "parameter_beam", seg, seg.track(pbeam))
            check("linac_two_cavities_off", "particle_beam", seg, seg.track(beam))
            res["linac_two_cavities_off"] = {"parameter_beam_us": timed_us(lambda: seg.track(pbeam)),
                                             "particle_beam_us": timed_us(lambda: seg.track(beam))}
            # a scan at C3's size — 4096 rows of settings (a quadrupole strength and six corrector angles per row, drawn on the host
            # from a fixed seed) over ONE shared beam of 1e5 particles, 11.5 GB of result — through six cells with one / six active
            # monitors: one stretch call (lattice_scan_wave_kernel). Checked against THIS engine's own track of single rows (the scan's
            # kernels are pinned to the reference and the oracle by tests/test_gpu_bigshape_stretch.py; the reference's float64 run of
            # the whole scan would be a 23 GB fixture): particles of rows 0 and 4095 bit for bit, readings to a float32 reading's rounding.
            import numpy as _np

            rng = _np.random.default_rng(20260930)
            B_scan = 4096
            k1_rows = torch.tensor(rng.standard_normal(B_scan), **kw)
            angle_rows = [torch.tensor(1e-5 * rng.standard_normal(B_scan), **kw) for _ in range(6)]

            def scan_lattice(monitors, row=None):
                pick = (lambda v: v) if row is None else (lambda v: v[row].clone())
                els = []
                for i in range(6):
                    els += [ca.Quadrupole(tt(0.2), k1=(pick(k1_rows) if i == 0 else tt(4.2 if i % 2 == 0 else -4.2)), **kw),
                            ca.HorizontalCorrector(tt(0.05), angle=pick(angle_rows[i]), **kw), ca.Drift(tt(0.8), **kw)]
                    if i % (6 // monitors) == 6 // monitors - 1:
                        els.append(ca.BPM(is_active=True, **kw))
                return ca.Segment(els)

            res["scan_4096_settings_x_1e5_shared_particles"] = {}
            for monitors in (1, 6):
                seg = scan_lattice(monitors)
                out = seg.track(beam)
                readings = torch.stack([e.reading for e in seg.elements if isinstance(e, ca.BPM)])          # (monitors, 4096, 2)
                size = float(beam.particles[:, [0, 2]].abs().max())
                for row in (0, B_scan - 1):
                    one = scan_lattice(monitors, row)
                    ref = one.track(beam)
                    if not torch.equal(out.particles[row], ref.particles):
                        raise AssertionError(f"DIAGNOSTICS_LATTICES scan with {monitors} monitors: row {row} differs from its own track")
                    r1 = torch.stack([e.reading for e in one.elements if isinstance(e, ca.BPM)])
                    err = float((readings[:, row].double() - r1.double()).abs().max()) / size
                    checks["values"] += 1
                    if not err < 2e-7:
                        raise AssertionError(f"DIAGNOSTICS_LATTICES scan with {monitors} monitors: readings of row {row} off by {err:.2e} beam sizes")
                del out
                us = timed_us(lambda: seg.track(beam), reps=5)
                res["scan_4096_settings_x_1e5_shared_particles"][f"{monitors}_monitor{'s' if monitors > 1 else ''}_ms"] = us / 1e3
                res["scan_4096_settings_x_1e5_shared_particles"][f"{monitors}_monitor{'s' if monitors > 1 else ''}_result_GBps"] = \
                    B_scan * beam.particles.shape[0] * 28 / (us * 1e-6) / 1e9
                checks["entries"] += 1
        res["checked_against_reference"] = checks
        return res

    def ares_speed_guard():
        # The reference's own speed guard (tests/test_speed.py:21-35): the ARES experimental-area section AREASOLA1 -> AREABSCR1 of the
        # lattice file with its screen switched on (method 'histogram', the file's default), 1e5 particles, `segment.track` +
        # `AREABSCR1.reading` — the reference asserts < 0.1 s. Checked against the reference's float64 run of the same beam
        # (tests/golden/ares_speed.json, generate_golden_ares_speed.py) before it is timed; no plan item may take the per-element
        # fallback, and the step's launches are counted with the profiler.
        import time as _t

        from benchmarks import diagnostics_inputs as di
        from cheetah_amd.accelerator import _planner

        want = json.load(open(os.path.join(ROOT, "tests", "golden", "ares_speed.json")))
        dt = torch.float32
        kw = {"dtype": dt, "device": device}
        full = ca.Segment.from_lattice_json(os.path.join(ROOT, "tests", "golden", "ares_lattice.json"), **kw)
        seg = full.subcell("AREASOLA1", "AREABSCR1")
        seg.AREABSCR1.is_active = True
        assert [type(e).__name__ for e in seg.elements] == want["elements"] and seg.AREABSCR1.method == want["method"]
        beam = ca.ParticleBeam(di.particles().to(device), torch.tensor(1e8, **kw), **kw)

        def step():
            out = seg.track(beam)
            return out, seg.AREABSCR1.reading

        with torch.no_grad():
            out, img = step()
            taken0 = dict(_planner.TAKEN)
            out, img = step()
            taken = {k: v - taken0[k] for k, v in _planner.TAKEN.items() if v != taken0[k]}
            h, w = img.shape
            got = {"sigma_x": float(out.sigma_x), "sigma_y": float(out.sigma_y), "image_sum": float(img.double().sum()),
                   "image_centre_x": float((img.double().sum(0) * torch.arange(w, dtype=torch.float64, device=device)).sum() / img.double().sum()),
                   "image_centre_y": float((img.double().sum(1) * torch.arange(h, dtype=torch.float64, device=device)).sum() / img.double().sum()),
                   "lit_pixels": int((img != 0).sum())}
            # float32 tracking against float64: 2e-4 of a beam size; a particle on a pixel's edge may fall to either side: the image's
            # sum exactly (the same particles inside the screen), its centre to a hundredth of a pixel, the lit pixels to 1 %
            assert [h, w] == want["image_shape"]
            for key, tol in (("sigma_x", 2e-4), ("sigma_y", 2e-4), ("image_sum", 1e-6), ("lit_pixels", 1e-2)):
                if not abs(got[key] / want[key] - 1.0) < tol:
                    raise AssertionError(f"ARES_EA_SPEED_GUARD: {key} = {got[key]!r}, the reference has {want[key]!r}")
            for key in ("image_centre_x", "image_centre_y"):
                if not abs(got[key] - want[key]) < 1e-2:
                    raise AssertionError(f"ARES_EA_SPEED_GUARD: {key} = {got[key]!r}, the reference has {want[key]!r}")
            if taken.get("element", 0) != 0:
                raise AssertionError(f"ARES_EA_SPEED_GUARD: {taken} — a plan item took the per-element fallback")
            launches = None
            try:
                from torch.profiler import ProfilerActivity, profile

                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    step()
                    torch.cuda.synchronize()
                launches = sum(1 for ev in prof.events() if ev.device_type is not None and "cuda" in str(ev.device_type).lower())
            except Exception:   # noqa: BLE001  (the count is a diagnostic; the timing below stands without it)
                launches = None
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            for _ in range(200):
                step()
            torch.cuda.synchronize()
            ms = (_t.perf_counter() - t0) / 200 * 1e3
        return {"workload": "ARES experimental area AREASOLA1 -> AREABSCR1 from tests/golden/ares_lattice.json, screen active (histogram), "
                            "1e5 particles, fp32: segment.track + AREABSCR1.reading (the reference's tests/test_speed.py:21-35, bound 100 ms)",
                "ms_per_step": ms, "reference_bound_ms": 100.0, "paths_taken": taken, "device_launches_per_step": launches,
                "checked_against_reference": {k: got[k] for k in got}, "particle_element_steps_per_s": 1e5 * len(seg.elements) / (ms * 1e-3)}

    for name, fn in (("C1", c1), ("C3", c3), ("C4", c4), ("C5", c5), ("DKD_FODO100", dkd), ("SECOND_ORDER_FODO100", second_order),
                     ("DIAGNOSTICS_LATTICES", diagnostics), ("ARES_EA_SPEED_GUARD", ares_speed_guard)):
        if only is None or name in only:
            guarded(name, fn)
    return out


def scaling_legs(ca, torch, dist, sharding, device, rank, world, steps, warmup) -> dict:
    """The other sharded workloads next to the weak-scaling headline. World > 1: RCCL over xGMI. World == 1: the same code
    on a one-rank RCCL group with `sharding.force_collectives` (every all-gather / all-reduce / chx_merge_moments executes;
    a one-rank exchange returns its input), so the legs are exercised by every single-GPU run of the driver."""
    from benchmarks import run_configs as rc

    rc.DEV = device
    dtype = torch.float32
    legs = {}
    # strong scaling of C2: 1e6 particles in TOTAL, split over the ranks
    seg = build_fodo(ca, torch, device, dtype)
    lo, hi = sharding.shard_range(N_PARTICLES, rank, world)
    torch.manual_seed(1234 + rank)
    beam = ca.ParticleBeam.from_parameters(num_particles=hi - lo, dtype=dtype, device=device)

    def strong():
        out = seg.track_elementwise(beam, fused=False)
        sharding.global_moments(out)

    def guarded(name, fn):
        """A leg that fails on this rank must not take the headline line down: the error is recorded; every rank reaches
        the barrier below either way (a failure inside a collective would still hang the others until the process-group
        timeout set in main())."""
        try:
            legs[name] = fn()
        except Exception as exc:
            legs[name] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.empty_cache()

    def strong_fused():
        out = seg.track_elementwise(beam, fused=True)      # the same maps, the same bits: every particle through the 100 maps in registers
        sharding.global_moments(out)

    def leg_strong():
        d = timed(torch, dist, strong, steps, warmup, world)
        res = {"scaling": "strong", "particles_total": N_PARTICLES, "particles_per_rank": hi - lo, "ms_per_step": d / steps * 1e3,
               "particle_element_steps_per_s": N_PARTICLES * len(seg.elements) * steps / d}
        # beside the 100 launches: ONE launch that keeps the particles in registers across the elements (chx_track_fused,
        # bit-identical results) — what a strong-scaling user would run once a rank's share is launch-bound (below ~5e5
        # particles a launch costs its ~3.7 us floor whatever it moves: benchmarks/strong_leg_trace.py)
        df = timed(torch, dist, strong_fused, steps, warmup, world)
        res["fused_in_register"] = {"ms_per_step": df / steps * 1e3, "particle_element_steps_per_s": N_PARTICLES * len(seg.elements) * steps / df}
        # (replaying the 100 apply launches from ONE device graph was measured and is not used: a replayed kernel node costs ~9 us
        # of scheduling on ROCm 7.2 against ~2.5 us for a launch issued from the C loop — 1.86 vs 1.03 ms at 1e6 particles, and
        # slower at the 1.25e5 particles a rank of an 8-GPU strong run holds as well: benchmarks/strong_leg_probe.py)
        return res

    guarded("c2_strong", leg_strong)

    def leg_c3():
        # C3: the 4096 settings split over the ranks, beam replicated, no collective
        B = 4096
        b0, b1 = sharding.shard_range(B, rank, world)
        k1 = torch.linspace(-30, 30, B, dtype=dtype, device=device)[b0:b1].contiguous()
        seg3 = rc.ares_subcell(dtype, k1)
        torch.manual_seed(99)
        beam3 = ca.ParticleBeam.from_parameters(num_particles=100_000, dtype=dtype, device=device)
        keep = {}
        d = timed(torch, dist, lambda: keep.__setitem__("o", seg3.track(beam3)), 5, 2, world)
        keep.clear()
        return {"scaling": "strong", "settings_total": B, "settings_per_rank": b1 - b0, "ms_per_track": d / 5 * 1e3,
                "particle_element_steps_per_s": B * 100_000 * 13 * 5 / d, "collectives": "none"}

    guarded("c3_batch_shard", leg_c3)

    def c4_lattice(n_local, total, seed):
        g = 128
        kw = {"dtype": dtype, "device": device}
        t = lambda v: torch.tensor(v, **kw)  # noqa: E731
        els = []
        for i in range(10):
            els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1)),
                    ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
        torch.manual_seed(seed)
        beam4 = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=n_local, total_charge=t(1e-9 * n_local / total),
                                                     energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3), radius_tau=t(1e-3),
                                                     sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
        return ca.Segment(els), beam4

    C4_COLLECTIVES = ("per kick: all-gather of 29 f64 per rank (beam moments; merged inside the geometry kernel) + all-reduce of the "
                      "8.4 MB charge grid (RCCL), between the two halves of the tile-ordered chain kick "
                      "(chx_sc_kick_sorted_begin / _finish)")

    def leg_c4():
        # C4 strong: 1e6 particles in TOTAL split over the ranks; every rank keeps its rows in deposit-tile order across the ten
        # kicks exactly as the un-sharded track does
        seg4, beam4 = c4_lattice(hi - lo, N_PARTICLES, 7 + rank)

        def c4():
            with sharding.particle_sharded():
                seg4.track(beam4)

        d = timed(torch, dist, c4, 20, 4, world)
        return {"scaling": "strong", "particles_total": N_PARTICLES, "particles_per_rank": hi - lo, "ms_per_track": d / 20 * 1e3,
                "particle_element_steps_per_s": N_PARTICLES * 50 * 20 / d, "collectives": C4_COLLECTIVES}

    def leg_c4_weak():
        # C4 weak: 1e6 particles PER RANK (the particle half of a kick — deposit + gather, ~95 us — stays what it is on one GPU,
        # the replicated Poisson solve and the two exchanges are the overhead that grows with the ranks)
        seg4, beam4 = c4_lattice(N_PARTICLES, N_PARTICLES * world, 70 + rank)

        def c4():
            with sharding.particle_sharded():
                seg4.track(beam4)

        d = timed(torch, dist, c4, 20, 4, world)     # (20 tracks: the 5 of earlier rounds scattered by +-0.1 ms from run to run)
        return {"scaling": "weak", "particles_per_rank": N_PARTICLES, "particles_total": N_PARTICLES * world,
                "ms_per_track": d / 20 * 1e3, "particle_element_steps_per_s": world * N_PARTICLES * 50 * 20 / d,
                "collectives": C4_COLLECTIVES}

    guarded("c4_particle_shard", leg_c4)
    guarded("c4_particle_shard_weak", leg_c4_weak)
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C1/C3/C4/C5 side timings (and the scaling legs)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1: open the one-rank RCCL group BEFORE the headline, so that its step runs the moments "
                         "all-gather + chx_merge_moments too (default: only the scaling legs, after the headline)")
    ap.add_argument("--no-scaling-legs", action="store_true", help="N = 1: skip the one-rank RCCL run of the scaling legs")
    ap.add_argument("--cpu-baseline-worker", type=str, default="", help=argparse.SUPPRESS)
    # smoke test of the multi-rank code path on a ONE-GPU box: every rank on cuda:0, collectives over gloo (RCCL refuses two
    # ranks on one device). Not a measurement.
    ap.add_argument("--one-device-gloo", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(args.cpu_baseline_worker)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)   # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    if args.one_device_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        # a rank that dies inside a collective must not park the others for the default 10 minutes
        if args.one_device_gloo:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=240))
        else:
            dist.init_process_group("nccl", device_id=torch.device(device), timeout=datetime.timedelta(seconds=240))
            assert dist.get_backend() == "nccl"
        assert dist.get_world_size() == args.gpus

    import cheetah_amd as ca
    from cheetah_amd import _ops, sharding

    def open_single_rank_group():
        """A process group of this one rank on RCCL + the force switch: every exchange takes its collective branch."""
        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device),
                                timeout=datetime.timedelta(seconds=240))
        assert dist.get_backend() == "nccl"
        sharding.force_collectives(True)

    if world == 1 and args.force_collectives:
        open_single_rank_group()

    dtype = torch.float32
    seg = build_fodo(ca, torch, device, dtype)
    E = len(seg.elements)
    torch.manual_seed(1234 + rank)
    beam = ca.ParticleBeam.from_parameters(num_particles=N_PARTICLES, dtype=dtype, device=device)

    state = {}

    def step_elementwise():
        out = seg.track_elementwise(beam, fused=False)
        state["moments"] = sharding.global_moments(out)
        state["out"] = out

    def step_fused():
        state["out"] = seg.track_elementwise(beam, fused=True)

    def step_merged():
        state["out"] = seg.track(beam)

    dt = timed(torch, dist, step_elementwise, args.steps, args.warmup, world)
    ms_per_step = dt / args.steps * 1e3
    value = world * N_PARTICLES * E * args.steps / dt
    sigma_x = float(state["moments"][8].sqrt())
    # the same step once more over a longer sample (>= 0.25 s of steps): the driver's --steps 20 is 19 ms of GPU work, and the
    # box-to-box scatter of so short a sample should be visible next to it. `steps` / `ms_per_step` above stay exactly as asked.
    long_steps = max(args.steps, int(math.ceil(250.0 / max(ms_per_step, 1e-3))))
    dt_long = timed(torch, dist, step_elementwise, long_steps, 0, world)
    headline_long = {"steps": long_steps, "ms_per_step": dt_long / long_steps * 1e3, "timed_region_s": dt_long,
                     "value": world * N_PARTICLES * E * long_steps / dt_long}

    def step_merged_no_grad():
        with torch.no_grad():
            state["out"] = seg.track(beam)

    modes = {}
    for name, fn in (("merged_reference_semantics", step_merged), ("merged_no_grad", step_merged_no_grad),
                     ("fused_in_register", step_fused)):
        d = timed(torch, dist, fn, args.steps, args.warmup, world)
        modes[name] = {"ms_per_track": d / args.steps * 1e3,
                       "particle_element_steps_per_s": world * N_PARTICLES * E * args.steps / d}
    modes["merged_no_grad"]["note"] = ("torch.no_grad(): without the requires_grad scan of the run's 300 setting tensors the "
                                       "merged track is bound by its two launches (~12 us of GPU)")

    # ---- small-beam workloads replayed from a device graph (rank 0 of a single-GPU run; each in a process of its own)
    if world == 1:
        modes["graph_replay"] = {"note": "torch.cuda.CUDAGraph (hipGraph) capture of the whole step, benchmarks/graph_modes.py: the "
                                         "README segment's track + screen reading, the control step with its five settings "
                                         "written in place (the replay follows them), a 16-cell linac with active cavities; "
                                         "eager times of the same process beside. control_assigned*: the README's own style, the "
                                         "five settings ASSIGNED as new tensors every step (eager only), beside the in-place step "
                                         "of the same process"}
        for which in ("c1", "control", "control_parameter_beam", "control_assigned", "control_assigned_parameter_beam", "linac"):
            try:
                proc = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "graph_modes.py"), which], capture_output=True,
                                      text=True, timeout=300, cwd=ROOT)
                line = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"graph_mode"')]
                if proc.returncode != 0 or not line:
                    raise RuntimeError(f"rc {proc.returncode}: {proc.stderr[-300:]}")
                modes["graph_replay"][which] = json.loads(line[-1])["graph_mode"]
            except Exception as exc:  # noqa: BLE001
                modes["graph_replay"][which] = {"error": str(exc)[:300]}

    # ---- roofline of the dominant kernel: HIP events on the launch stream around the E-launch run, every step
    ms_run = event_timed_elementwise(torch, seg, beam, args.steps, args.warmup)
    ms_launch = ms_run / E
    algo_bytes = 56.0 * N_PARTICLES  # 7 fp32 read + 7 fp32 written per particle per launch
    achieved = algo_bytes / (ms_launch * 1e-3) / 1e9
    traffic = traffic_big = None
    tpath = os.path.join(ROOT, "profiles", "apply_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_big = (tj.get("streaming") or {}).get("hbm_bytes_per_launch") or None
        except Exception:
            traffic = None
    traffic_source = "profiles/apply_traffic.json (this round's rocprofv3 --pmc passes, builder session)"
    if world == 1 and rank == 0:
        live = live_pmc_traffic()
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_big = live.get("streaming_hbm_bytes_per_launch", traffic_big)
            traffic_source = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                              "profiles/traffic_probe.py in subprocesses; KiB units, FETCH x 2 (gfx950)")
    ms_rocprof = rocprof_average_ms()
    # The live number is the wall time of the 100 back-to-back launches / 100. rocprofv3's kernel trace brackets every
    # dispatch from its first wave to its last, and consecutive dispatches of one stream overlap by a few hundred ns, so its
    # mean duration is slightly LONGER than the per-launch wall time. The workload-size figure uses the larger of the two (the
    # conservative reading, identical to what the tracked profile gives); both durations are reported.
    ms_used = max(ms_launch, ms_rocprof) if ms_rocprof else ms_launch
    # Top level = the kernel the METRIC times, at the workload's own launch shape: apply_tile_kernel<float,2,0> on 1e6 particles.
    # Its 28 MB in + 28 MB out per launch live in the 256 MiB Infinity Cache between launches, so the bound is labelled
    # "infinity-cache" and `frac` (against the 8 TB/s HBM spec, as the contract asks) is not an HBM rate; `hbm_streaming` is the
    # same arithmetic streaming from HBM proper (1.6e7 particles, 448 MB in + 448 MB out per launch: apply_wave_kernel).
    roofline = {"bound": "infinity-cache", "kernel": APPLY_KERNEL, "particles": N_PARTICLES,
                "achieved": algo_bytes / (ms_used * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": algo_bytes / (ms_used * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": ms_used, "launches_per_step": E,
                "avg_launch_ms_events": ms_launch, "avg_launch_ms_rocprof": ms_rocprof,
                "rocprof_summary": os.path.relpath(PROFILE_CSV, ROOT),
                "note": "56 B x 1e6 particles per launch / the LARGER of the live HIP-event average (100 back-to-back launches / "
                        "100) and the rocprofv3 kernel-trace average of the tracked summary; working set 56 MB = Infinity-Cache "
                        "resident, `traffic` (PMC, HBM side) is therefore below the algorithmic bytes"}
    big_n = 16_000_000
    stream = {"bound": "hbm", "kernel": "apply_wave_kernel<float, 1, 64>", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": None, "traffic": None}
    try:
        big = ca.ParticleBeam.from_parameters(num_particles=big_n, dtype=dtype, device=device)
        seg10 = ca.Segment(list(seg.elements)[:10])
        ms_big = event_timed_elementwise(torch, seg10, big, 10, 2) / 10
        gbs = 56.0 * big_n / (ms_big * 1e-3) / 1e9
        stream.update({"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "frac_vs_copy_ceiling": gbs / COPY_CEILING_GBS,
                       "copy_ceiling": COPY_CEILING_GBS, "particles": big_n, "algorithmic_bytes_per_launch": 56.0 * big_n,
                       "avg_launch_ms": ms_big, "traffic": traffic_big,
                       "traffic_note": "PMC FETCH_SIZE x2 + WRITE_SIZE per launch at this size (separate rocprofv3 --pmc passes, "
                                       "gfx950 corrections of MI355X_MICROARCH.md; profiles/r06_pmc_apply.md): 1.0002 x the "
                                       "algorithmic bytes"})
        del big, seg10
        torch.cuda.empty_cache()
    except Exception as exc:
        stream["error"] = f"{type(exc).__name__}: {exc}"
    roofline["hbm_streaming"] = stream

    # the tracked beam itself is checked, not only timed: sigma_x of the outgoing beam of rank 0's seed against the committed value
    # (tests/test_gpu_bench_parity.py holds the bit-exact comparison of this very step with the oracle)
    # (N > 1: the moments are those of the union of the ranks' beams — other seeds, the same distribution: 5e-3)
    sigma_ok = abs(sigma_x / EXPECTED_SIGMA_X_RANK0 - 1.0) < (1e-9 if world == 1 else 5e-3) if rank == 0 else None
    if rank == 0 and not sigma_ok:
        raise SystemExit(f"bench.py: sigma_x of the tracked beam is {sigma_x!r}, expected {EXPECTED_SIGMA_X_RANK0!r}: the step "
                         "does not compute what it is timed for")

    result = {
        "metric": "particle-element-steps/sec at 1e6 particles, 100-elem linac",
        "value": value, "unit": "particle-element-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: 100-element Drift+Quadrupole FODO, 1e6 particles per GPU, fp32, "
                               "element-by-element tracking (no map merging) + global beam moments",
                   "elements": E, "particles_per_gpu": N_PARTICLES, "parallelism": f"particle-shard x{world}",
                   "sigma_x_out": sigma_x, "sigma_x_expected": EXPECTED_SIGMA_X_RANK0, "sigma_x_checked": sigma_ok},
        "timed_region_s": dt, "headline_long": headline_long, "modes": modes, "roofline": roofline,
    }
    state.clear()
    torch.cuda.empty_cache()
    if not args.no_configs:
        if world == 1:
            result["configs"] = other_configs(ca, torch, device)
        if world > 1 or not args.no_scaling_legs:
            try:
                if world == 1 and not dist.is_initialized():
                    open_single_rank_group()
                legs = scaling_legs(ca, torch, dist, sharding, device, rank, world, min(args.steps, 50), min(args.warmup, 5))
                legs["n_gpus"] = world
                if world == 1:
                    legs["note"] = ("one-rank RCCL group with forced collectives: all_gather_into_tensor -> chx_merge_moments "
                                    "and the device all-reduce of the charge grid execute; not a scaling measurement")
                result["scaling_legs"] = legs
            except Exception as exc:   # the headline stands on its own
                result["scaling_legs"] = {"n_gpus": world, "error": f"{type(exc).__name__}: {exc}"}
    result["collectives_forced_in_headline"] = bool(world == 1 and args.force_collectives)
    # who ran: the process group as torch.distributed sees it and the device every rank computed on (for the driver to check that
    # RCCL saw N ranks on N distinct GPUs)
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": torch.cuda.current_device(),
          "device_name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    result["distributed"] = {"initialized": bool(dist.is_available() and dist.is_initialized()),
                             "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                             "backend": dist.get_backend() if dist.is_initialized() else None,
                             "distinct_devices": len({(r["device_index"], r["pci_bus_id"], r["uuid"]) for r in ranks}), "ranks": ranks}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(E)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The ONE JSON line is the last thing on stdout: whatever native libraries still hold in C's stdio buffer (RCCL prints its
        # library path there; into a pipe that buffer is only written out at process exit, i.e. behind Python's own) goes out first.
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

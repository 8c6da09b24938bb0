#!/usr/bin/env python3
"""bench.py — headline benchmark: particle-element-steps/s, 100-element FODO linac, 1e6 particles / GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], "C2"): 25 x [Quad(0.2, +4.2), Drift(0.8), Quad(0.2, -4.2), Drift(0.8)],
ParticleBeam.from_parameters defaults, fp32, 1e6 particles PER RANK (weak scaling: the particle axis
shards with no data-path collective; only the global beam moments are all-reduced over RCCL).

One timed "step" = `Segment.track_elementwise(beam)` — all 100 elements applied one after the other,
one pass over HBM per element, NO map merging — followed by the global beam moments of the outgoing
beam (2 reduction passes + 2 tiny all-reduces).  `value` = N_gpus * 1e6 * 100 / t_step.
The reference's own semantics (`Segment.track`: merge the 100 maps, one pass) and the fused
in-register variant are timed as well and reported under "modes" — N*E/t is not a bandwidth
measure for those ("The metric is ill-posed under matrix merging").

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_PARTICLES = 1_000_000
N_CELLS = 25
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def build_fodo(ca, torch, device, dtype):
    t = lambda v: torch.tensor(v, dtype=dtype, device=device)  # noqa: E731
    kw = {"dtype": dtype, "device": device}
    els = []
    for _ in range(N_CELLS):
        els += [ca.Quadrupole(t(0.2), k1=t(4.2), **kw), ca.Drift(t(0.8), **kw),
                ca.Quadrupole(t(0.2), k1=t(-4.2), **kw), ca.Drift(t(0.8), **kw)]
    return ca.Segment(els)


def timed(torch, dist, fn, steps, warmup, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def cpu_baseline(n_elements):
    """The CPU oracle (C restatement of the reference path, OpenMP over all host cores) on the same
    workload and mode: element-by-element fp32 tracking of 1e6 particles, repeated until >= ~10 s."""
    import numpy as np

    from oracle import chx_oracle as oracle

    E = 1e8
    f = np.float32
    cell = [oracle.build_rmatrix("quadrupole", [f(0.2), f(4.2), 0, 0, 0], E).astype(np.float32),
            oracle.build_rmatrix("drift", [f(0.8)], E).astype(np.float32),
            oracle.build_rmatrix("quadrupole", [f(0.2), f(-4.2), 0, 0, 0], E).astype(np.float32),
            oracle.build_rmatrix("drift", [f(0.8)], E).astype(np.float32)]
    maps = np.concatenate((cell * N_CELLS)[:n_elements])
    rng = np.random.default_rng(1234)
    x = (rng.standard_normal((N_PARTICLES, 7)) * [175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3, 0]).astype(np.float32)
    x[..., 6] = 1.0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    out, tmp, x_par = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    # first touch of every buffer happens inside the OpenMP loops (NUMA-local pages on a multi-socket host)
    oracle.track_elementwise(x, np.eye(7, dtype=np.float32)[None], x_par, tmp)
    x = x_par
    oracle.track_elementwise(x, maps, out, tmp)  # warm-up: page faults, OpenMP pool
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.track_elementwise(x, maps, out, tmp)
        reps += 1
        el = time.perf_counter() - t0
        if el >= 10.0 or reps >= 400:
            break
    return {"value": N_PARTICLES * len(maps) * reps / el, "unit": "particle-element-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{reps} x (1e6 particles x {len(maps)} elements, element-by-element, fp32 fma chain, "
                      f"C oracle with OpenMP on {cores} host threads), {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE is {world})")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(device))

    import cheetah_amd as ca
    from cheetah_amd import _ops, sharding

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

    modes = {}
    for name, fn in (("merged_reference_semantics", step_merged), ("fused_in_register", step_fused)):
        d = timed(torch, dist, fn, args.steps, args.warmup, world)
        modes[name] = {"ms_per_track": d / args.steps * 1e3,
                       "particle_element_steps_per_s": world * N_PARTICLES * E * args.steps / d}

    # roofline of the dominant kernel (apply_tile_kernel<float>): HIP events on the launch stream
    R = seg.elements[0].first_order_transfer_map(beam.energy, beam.species).reshape(1, 7, 7).contiguous()
    x = beam.particles.reshape(1, N_PARTICLES, 7)
    scratch = torch.empty_like(x)
    ms_launch = _ops.time_apply_ms(x, R, scratch, 1, 1, 1, N_PARTICLES, 200)
    algo_bytes = 56.0 * N_PARTICLES  # 7 fp32 read + 7 fp32 written per particle per launch
    achieved = algo_bytes / (ms_launch * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "apply_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "apply_tile_kernel<float,2,0>", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": ms_launch,
                "note": "28 MB in + 28 MB out per launch fit the 256 MiB Infinity Cache; see DESIGN.md"}

    result = {
        "metric": "particle-element-steps/sec at 1e6 particles, 100-elem linac",
        "value": value, "unit": "particle-element-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: 100-element Drift+Quadrupole FODO, 1e6 particles per GPU, fp32, "
                               "element-by-element tracking (no map merging) + global beam moments",
                   "elements": E, "particles_per_gpu": N_PARTICLES, "parallelism": f"particle-shard x{world}",
                   "sigma_x_out": sigma_x},
        "modes": modes, "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(E)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

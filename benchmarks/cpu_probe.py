"""Host probe for the cpu_baseline leg of bench.py: what the GPU box's host offers (cores, affinity, NUMA) and how the two CPU
restatements scale with threads and binding. Prints one JSON object per line."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as exc:  # noqa: BLE001
        return f"failed: {exc}"


def worker(kind, threads):
    import numpy as np

    N, E = 1_000_000, 100
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((N, 7)) * [175e-6, 4e-6, 175e-6, 4e-6, 8e-6, 2e-3, 0]).astype(np.float32)
    x[:, 6] = 1
    R = np.eye(7, dtype=np.float32)
    R[0, 1] = 0.8
    maps = np.stack([R] * E)
    if kind == "oracle":
        from oracle import chx_oracle as oracle

        out, tmp, xs = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        oracle.track_elementwise(x, np.eye(7, dtype=np.float32)[None], xs, tmp)
        oracle.track_elementwise(xs, maps, out, tmp)
        t0, reps = time.perf_counter(), 0
        while time.perf_counter() - t0 < 2.0:
            oracle.track_elementwise(xs, maps, out, tmp)
            reps += 1
        el = time.perf_counter() - t0
    else:
        import torch

        torch.set_num_threads(threads)
        xt = torch.from_numpy(x)
        mt = [torch.from_numpy(m) for m in maps]
        with torch.no_grad():
            y = xt
            for m in mt[:8]:
                y = y @ m.mT
            t0, reps = time.perf_counter(), 0
            while time.perf_counter() - t0 < 2.0:
                if kind == "torch":
                    y = xt
                    for m in mt:
                        y = y @ m.mT
                else:                                # ping-pong buffers: no allocation in the loop
                    a, b = torch.empty_like(xt), torch.empty_like(xt)
                    torch.matmul(xt, mt[0].mT, out=a)
                    for m in mt[1:]:
                        torch.matmul(a, m.mT, out=b)
                        a, b = b, a
                reps += 1
            el = time.perf_counter() - t0
    print(json.dumps({"kind": kind, "threads": threads, "bind": os.environ.get("OMP_PROC_BIND"), "places": os.environ.get("OMP_PLACES"),
                      "steps_per_s": N * E * reps / el, "ms_per_pass": el / reps / E * 1e3}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        worker(sys.argv[1], int(sys.argv[2]))
        sys.exit(0)
    print(json.dumps({"nproc": sh("nproc"), "nproc_all": sh("nproc --all"), "affinity": sh("taskset -p $$"),
                      "lscpu": sh("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|L3|^CPU\\(s\\)'"),
                      "cgroup_cpu": sh("cat /sys/fs/cgroup/cpu.max 2>/dev/null"), "numactl": sh("numactl -H 2>/dev/null | head -20"),
                      "mem": sh("free -g | head -2")}))
    import psutil

    phys = psutil.cpu_count(logical=False)
    avail = len(os.sched_getaffinity(0))
    print(json.dumps({"physical": phys, "logical": psutil.cpu_count(), "affinity_count": avail}))
    for kind in ("oracle", "torch", "torch_out"):
        for threads in sorted({1, 4, 8, 16, 32, 64, min(phys, avail), avail}):
            if threads > avail:
                continue
            for bind in (None, ("spread", "cores"), ("close", "cores")):
                if kind != "oracle" and bind is not None and bind[0] == "close":
                    continue
                env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="",
                           CUDA_VISIBLE_DEVICES="")
                env.pop("OMP_PROC_BIND", None)
                env.pop("OMP_PLACES", None)
                if bind:
                    env["OMP_PROC_BIND"], env["OMP_PLACES"] = bind
                r = subprocess.run([sys.executable, os.path.abspath(__file__), kind, str(threads)], env=env, capture_output=True,
                                   text=True, timeout=120)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                print(line[-1] if line else json.dumps({"kind": kind, "threads": threads, "error": r.stderr[-200:]}), flush=True)

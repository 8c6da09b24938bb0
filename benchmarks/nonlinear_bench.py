"""Throughput of the non-linear tracking kernels (chx_dkd_track, chx_apply_second_order) on one GPU.

    python benchmarks/nonlinear_bench.py [N]

Per kernel: average launch time over 50 back-to-back launches (torch.cuda events on the launch stream), the
algorithmic HBM rate (7 values read + 7 written per particle) and particles/s.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cheetah_amd import _ops  # noqa: E402


def timed(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    torch.manual_seed(0)
    for dt in (torch.float32, torch.float64):
        esz = 4 if dt == torch.float32 else 8
        x = torch.randn(N, 7, device="cuda", dtype=dt) * torch.tensor([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0], device="cuda", dtype=dt)
        x[:, 6] = 1.0
        E = torch.tensor(1e8, device="cuda", dtype=dt)
        m, nq = 510998.95069, -1.0
        t = lambda *v: torch.tensor([list(v)], device="cuda", dtype=dt)  # noqa: E731
        cases = {
            "apply_affine7 (linear)": lambda: _ops.apply_map(x, R),
            "dkd drift": lambda: _ops.dkd_track(0, x, t(1.0), torch.Size(()), E, m, nq),
            "dkd quadrupole 1 step": lambda: _ops.dkd_track(1, x, t(0.2, 4.2, 0.0, 0.0, 0.0), torch.Size(()), E, m, nq, 1),
            "dkd quadrupole 10 steps": lambda: _ops.dkd_track(1, x, t(0.2, 4.2, 0.1, 1e-4, -1e-4), torch.Size(()), E, m, nq, 10),
            "dkd dipole": lambda: _ops.dkd_track(2, x, t(0.5, 0.35, 0.17, 0.17, 0.1, 0.5, 0.5, 0.05, 0.05), torch.Size(()), E, m, nq, 1, 3),
            "dkd tdc": lambda: _ops.dkd_track(3, x, t(1.0, 1e7, 0.2, 1e9, 0.0, 0.0, 0.0), torch.Size(()), E, m, nq),
            "second_order apply": lambda: _ops.apply_second_order(x, T),
        }
        if dt == torch.float32:   # the float32-arithmetic variants (Element.dkd_precision = "storage")
            cases.update({
                "dkd drift, fp32 arith": lambda: _ops.dkd_track(0, x, t(1.0), torch.Size(()), E, m, nq, storage_precision=True),
                "dkd quadrupole 1 step, fp32 arith": lambda: _ops.dkd_track(1, x, t(0.2, 4.2, 0.0, 0.0, 0.0), torch.Size(()), E, m, nq, 1,
                                                                            storage_precision=True),
                "dkd quadrupole 10 steps, fp32 arith": lambda: _ops.dkd_track(1, x, t(0.2, 4.2, 0.1, 1e-4, -1e-4), torch.Size(()), E, m, nq,
                                                                              10, storage_precision=True),
                "dkd dipole, fp32 arith": lambda: _ops.dkd_track(2, x, t(0.5, 0.35, 0.17, 0.17, 0.1, 0.5, 0.5, 0.05, 0.05), torch.Size(()), E,
                                                                 m, nq, 1, 3, storage_precision=True),
                "dkd tdc, fp32 arith": lambda: _ops.dkd_track(3, x, t(1.0, 1e7, 0.2, 1e9, 0.0, 0.0, 0.0), torch.Size(()), E, m, nq,
                                                              storage_precision=True),
            })
        R = torch.eye(7, device="cuda", dtype=dt)
        R[0, 1] = 1.0
        T = _ops.build_ttensor(1, t(0.2, 4.2, 0.1, 1e-4, -1e-4), torch.Size(()), E, m)
        print(f"--- N = {N:.0e}, {dt}")
        for name, fn in cases.items():
            us = timed(fn)
            gbs = N * 14 * esz / us * 1e-3
            print(f"{name:36s} {us:9.1f} us   {gbs:8.1f} GB/s   {N / us:8.1f} Mparticles/s")
        if dt == torch.float32:
            # error of the two arithmetic widths against the float64 kernel on the same (float32-representable) input, in units of
            # the largest absolute value of each coordinate
            x64, E64 = x.double(), E.double()
            t64 = lambda *v: torch.tensor([list(v)], device="cuda", dtype=torch.float64)  # noqa: E731
            for name, kind, par, extra in (("drift", 0, (1.0,), ()), ("quadrupole", 1, (0.2, 4.2, 0.1, 1e-4, -1e-4), (10,)),
                                           ("dipole", 2, (0.5, 0.35, 0.17, 0.17, 0.1, 0.5, 0.5, 0.05, 0.05), (1, 3)),
                                           ("tdc", 3, (1.0, 1e7, 0.2, 1e9, 0.0, 0.0, 0.0), ())):
                ref = _ops.dkd_track(kind, x64, t64(*par), torch.Size(()), E64, m, nq, *extra)[0]
                scale = ref.abs().max(dim=0).values[:6]
                for label, flag in (("fp64 arith", False), ("fp32 arith", True)):
                    got = _ops.dkd_track(kind, x, t(*par), torch.Size(()), E, m, nq, *extra, storage_precision=flag)[0]
                    err = ((got.double() - ref).abs().max(dim=0).values[:6] / scale).tolist()
                    print(f"error vs float64 kernel, {name:10s} {label}: " + " ".join(f"{v:.1e}" for v in err))
        us = timed(lambda: _ops.build_ttensor(1, t(0.2, 4.2, 0.1, 1e-4, -1e-4), torch.Size(()), E, m))
        print(f"{'build_ttensor (1 row)':28s} {us:9.1f} us")


if __name__ == "__main__":
    main()

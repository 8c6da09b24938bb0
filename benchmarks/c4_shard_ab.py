#!/usr/bin/env python3
"""C4 with its particles on a forced one-rank RCCL group (bench.py `scaling_legs.c4_particle_shard`), 40 tracks: what the exchanges
and the launches around them cost on one GPU. Run with CHX_SC_RIDERS=0 / 1 (the bookkeeping step as a launch of its own / riding in the
convolution's first pass) to compare."""
import datetime
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import sharding  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if "MASTER_PORT" not in os.environ:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"), timeout=datetime.timedelta(seconds=240))
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)  # noqa: E731
els = []
for i in range(10):
    els += [ca.Drift(t(0.1), **kw), ca.SpaceChargeKick(t(0.2), grid_shape=(128, 128, 128), **kw), ca.Drift(t(0.1), **kw),
            ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1), **kw)]
seg = ca.Segment(els)
torch.manual_seed(7)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3),
                                            radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)


def timed(fn, reps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def sharded():
    with sharding.particle_sharded(force_collectives=True):
        seg.track(beam)


print("riders", os.environ.get("CHX_SC_RIDERS", "1"), " un-sharded %.3f ms" % timed(lambda: seg.track(beam), 40, 6), " forced one-rank group %.3f ms" % timed(sharded, 40, 6))
dist.destroy_process_group()

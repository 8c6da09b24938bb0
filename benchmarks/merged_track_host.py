#!/usr/bin/env python3
"""Where the wall time of a merged `Segment.track` goes (C2 lattice: 100 elements, 1e6 particles, fp32) — and what a
captured hipGraph would change. Lines printed (microseconds per track, 2000 tracks each):
  Segment.track               the product path: persistent run plan, one C call (chx_run_track), new ParticleBeam
  Segment.track, no_grad      the same without the requires_grad scan of the run's tensors
  chx_run_track only          the C call alone with pre-made arguments (two kernel launches), no Python objects
  chx_apply_affine7 only      one launch with a stored map
  hipGraph replay             torch.cuda.CUDAGraph capture of the chx_run_track call on fixed buffers
  GPU time per track          hipEvent time of 2000 back-to-back chx_run_track calls / 2000
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca  # noqa: E402
from cheetah_amd import _lib, _ops  # noqa: E402

import bench  # noqa: E402

dt = torch.float32
seg = bench.build_fodo(ca, torch, "cuda", dt)
beam = ca.ParticleBeam.from_parameters(num_particles=1_000_000, dtype=dt, device="cuda")
REPS = 2000


def wall(fn, reps=REPS):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print(f"Segment.track               {wall(lambda: seg.track(beam)):7.1f} us")
with torch.no_grad():
    print(f"Segment.track, no_grad      {wall(lambda: seg.track(beam)):7.1f} us")
fr = seg._plan()[0][1].fast
assert fr is not None and fr.ok
lib = _lib.lib()
x = beam.particles
out = torch.empty_like(x)
sp = beam.species
args = (fr.kinds, fr.ptrs, fr.E, beam.energy.data_ptr(), sp.mass_eV_float, sp.num_elementary_charges_float, fr.code,
        fr.state.data_ptr(), fr.state_bytes, x.data_ptr(), out.data_ptr(), x.shape[0], None, None)
print(f"chx_run_track only          {wall(lambda: lib.chx_run_track(*args, _ops.stream_ptr())):7.1f} us")
R = seg.first_order_transfer_map(beam.energy, beam.species).contiguous()
print(f"chx_apply_affine7 only      "
      f"{wall(lambda: lib.chx_apply_affine7(x.data_ptr(), R.data_ptr(), out.data_ptr(), 1, 1, 1, x.shape[0], 0, _ops.stream_ptr())):7.1f} us")
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    lib.chx_run_track(*args, _ops.stream_ptr())
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        lib.chx_run_track(*args, _ops.stream_ptr())
torch.cuda.current_stream().wait_stream(side)
print(f"hipGraph replay             {wall(graph.replay):7.1f} us")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    lib.chx_run_track(*args, _ops.stream_ptr())
e1.record()
torch.cuda.synchronize()
print(f"GPU time per track          {e0.elapsed_time(e1) / REPS * 1e3:7.1f} us")

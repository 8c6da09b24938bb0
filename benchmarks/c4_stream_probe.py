import os, sys, time, torch
sys.path.insert(0, '/root/repo')
import cheetah_amd as ca
from cheetah_amd.accelerator.space_charge_kick import SpaceChargeKick
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
g = 128
els = []
for i in range(10):
    els += [ca.Drift(t(0.1)), ca.SpaceChargeKick(t(0.2), grid_shape=(g, g, g), **kw), ca.Drift(t(0.1)),
            ca.Quadrupole(t(0.1), k1=t(4.2 if i % 2 == 0 else -4.2), **kw), ca.Drift(t(0.1))]
seg = ca.Segment(els)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=1_000_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3), radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
def bench(label):
    for _ in range(3): seg.track(beam)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): seg.track(beam)
    torch.cuda.synchronize(); print(label, "ms/track", round((time.perf_counter() - t0) * 100, 3))
bench("two streams")
orig = SpaceChargeKick._side_stream
SpaceChargeKick._side_stream = classmethod(lambda cls, device: None)
bench("one stream")

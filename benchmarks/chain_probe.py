import sys, torch
sys.path.insert(0, '/root/repo')
import cheetah_amd as ca
from cheetah_amd import _ops
from benchmarks import run_configs as rc
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
orig = _ops.sc_kick_sorted
log = []
def spy(*a, **k):
    r = orig(*a, **k)
    h = a[9][:32].view(torch.int32).tolist()
    log.append(h)
    return r
_ops.sc_kick_sorted = spy
out = seg.track(beam)
for h in log: print(h)
print("sigma", float(out.sigma_x), float(out.sigma_y), float(out.sigma_tau))

"""C4 with the Green chain on the MAIN stream (no overlap): run under `rocprofv3 --kernel-trace` for the un-inflated duration of
every kernel of a chain kick; prints the crosser counts of the chain's kicks."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cheetah_amd as ca
from cheetah_amd import _ops
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
if "--two" not in sys.argv:
    SpaceChargeKick._side_stream = classmethod(lambda cls, device: None)
orig = _ops.sc_kick_sorted
log = []
def spy(*a, **k):
    r = orig(*a, **k)
    log.append(a[9])
    return r
_ops.sc_kick_sorted = spy
for _ in range(3):
    log.clear()
    seg.track(beam)
    torch.cuda.synchronize()
print("header (parity, scatter_now, misfiled, sum of misfiled permille, misfiled, n_sorts, n_deposits) after the track:",
      log[-1][:32].view(torch.int32).tolist()[:7])

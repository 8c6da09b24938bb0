import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
kw = {"dtype": torch.float32, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)
beam = ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=10_000, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3),
                                            radius_tau=t(1e-3), sigma_px=t(1e-6), sigma_py=t(1e-6), sigma_p=t(1e-6), **kw)
kick = ca.SpaceChargeKick(t(0.2), grid_shape=(32, 32, 32), **kw)
with torch.no_grad():
    for _ in range(6):
        kick.track(beam)
    torch.cuda.synchronize()

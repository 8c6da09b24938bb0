#!/usr/bin/env python3
"""ParticleBeam / ParameterBeam constructors and utilities at 1e6 particles: ms per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cheetah_amd as ca
dt = torch.float32
kw = {"dtype": dt, "device": "cuda"}
t = lambda v: torch.tensor(v, **kw)


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


N = 1_000_000
beam = ca.ParticleBeam.from_parameters(num_particles=N, **kw)
pb = ca.ParameterBeam.from_parameters(**kw)
cases = {
    "ParticleBeam.from_parameters": lambda: ca.ParticleBeam.from_parameters(num_particles=N, **kw),
    "ParticleBeam.from_twiss": lambda: ca.ParticleBeam.from_twiss(beta_x=t(3.0), beta_y=t(5.0), num_particles=N, **kw),
    "ParticleBeam.uniform_3d_ellipsoid": lambda: ca.ParticleBeam.uniform_3d_ellipsoid(num_particles=N, **kw),
    "ParticleBeam.make_linspaced": lambda: ca.ParticleBeam.make_linspaced(num_particles=N, **kw),
    "beam.transformed_to(sigma_x=...)": lambda: beam.transformed_to(sigma_x=t(2e-4), mu_y=t(1e-4)),
    "beam.as_parameter_beam()": lambda: beam.as_parameter_beam(),
    "pb.as_particle_beam(N)": lambda: pb.as_particle_beam(num_particles=N),
    "all 12 first / second moments": lambda: [getattr(beam, n) for n in ("mu_x", "mu_px", "mu_y", "mu_py", "mu_tau", "mu_p", "sigma_x", "sigma_px", "sigma_y", "sigma_py", "sigma_tau", "sigma_p")],
    "emittances, twiss": lambda: [getattr(beam, n) for n in ("emittance_x", "emittance_y", "beta_x", "beta_y", "alpha_x", "alpha_y", "normalized_emittance_x")],
    "beam.clone()": lambda: beam.clone(),
    "beam.energies / momenta": lambda: (beam.energies, beam.momenta),
    "beam.to_xyz_pxpypz()": lambda: beam.to_xyz_pxpypz(),
    "beam.linspaced(1000)": lambda: beam.linspaced(1000),
}
with torch.no_grad():
    for name, fn in cases.items():
        try:
            print(f"{name:40s}: {timeit(fn):9.3f} ms", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{name:40s}: {type(exc).__name__}: {str(exc)[:100]}", flush=True)

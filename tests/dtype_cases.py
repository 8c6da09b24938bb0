"""The dtype sweep (tests/golden/dtypes.json): every element kind built in float32 or float64, tracked with a float32 or float64
beam — does it raise, and what dtype comes out? The same code runs against the reference (generator) and this engine (test)."""
import torch

DT = {"f32": torch.float32, "f64": torch.float64}


def elements(module, dt, dev):
    kw = {"dtype": dt, "device": dev}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    return {
        "Drift": lambda: module.Drift(t(0.5), **kw),
        "Quadrupole": lambda: module.Quadrupole(t(0.2), k1=t(3.0), **kw),
        "Dipole": lambda: module.Dipole(t(0.5), angle=t(0.1), **kw),
        "HorizontalCorrector": lambda: module.HorizontalCorrector(t(0.1), angle=t(1e-4), **kw),
        "Solenoid": lambda: module.Solenoid(t(0.2), k=t(1.0), **kw),
        "Cavity": lambda: module.Cavity(t(0.5), voltage=t(5e6), phase=t(10.0), frequency=t(1.3e9), **kw),
        "Marker": lambda: module.Marker(**kw),
        "BPM": lambda: module.BPM(is_active=True, **kw),
        "Aperture": lambda: module.Aperture(x_max=t(1e-3), y_max=t(1e-3), is_active=True, **kw),
        "Screen": lambda: module.Screen(resolution=(40, 30), pixel_size=t([5e-5, 5e-5]), is_active=True, **kw),
        "SpaceChargeKick": lambda: module.SpaceChargeKick(t(0.1), grid_shape=(8, 8, 8), **kw),
        "Quadrupole_second_order": lambda: module.Quadrupole(t(0.2), k1=t(3.0), tracking_method="second_order", **kw),
        "Quadrupole_drift_kick_drift": lambda: module.Quadrupole(t(0.2), k1=t(3.0), tracking_method="drift_kick_drift", **kw),
        "Segment": lambda: module.Segment([module.Drift(t(0.5), **kw), module.Quadrupole(t(0.2), k1=t(3.0), **kw), module.Drift(t(0.5), **kw)]),
    }


def beam(module, kind, dt, dev):
    kw = {"dtype": dt, "device": dev}
    n = 64
    x = torch.stack([torch.linspace(-1, 1, n, **kw) * s for s in (2e-4, 3e-5, -2e-4, 2e-5, 1e-4, 1e-3)] + [torch.ones(n, **kw)], dim=-1)
    x[:, 2] = x[:, 2].roll(7)
    x[:, 4] = x[:, 4].roll(13)
    if kind == "particle":
        return module.ParticleBeam(x, torch.tensor(1e8, **kw), particle_charges=torch.full((n,), 1e-14, **kw), **kw)
    cov = torch.zeros(7, 7, **kw)
    cov[:6, :6] = torch.cov(x[:, :6].T)
    mu = torch.cat([x.mean(dim=0)[:6] + 1e-4, torch.ones(1, **kw)])
    return module.ParameterBeam(mu, cov, torch.tensor(1e8, **kw), **kw)


def outcome(module, name, e_dt, b_dt, kind, dev):
    el = elements(module, DT[e_dt], dev)[name]()
    b = beam(module, kind, DT[b_dt], dev)
    out = el.track(b)
    ref = out.particles if kind == "particle" else out.mu
    d = {"dtype": str(ref.dtype), "energy_dtype": str(out.energy.dtype)}
    if name == "Screen":
        d["reading_dtype"] = str(el.reading.dtype)
    if name == "BPM":
        d["reading_dtype"] = str(el.reading.dtype)
    return d

"""The inputs of the broadcasting sweep (tests/golden/shapes.json), built from shapes alone by the same formula for the
reference (in tests/golden/generate_golden_shapes.py) and for this engine (tests/test_gpu_shapes.py): ramps, no random draws."""
import torch

SHAPES = [(), (1,), (3,), (1, 3), (2, 1)]


def inputs(module, k1_shape, beam_shape, energy_shape, kind, lattice, dev=None):
    """The same numbers on both sides: ramps, no random draws."""
    kw = {"dtype": torch.float64, "device": dev}
    ramp = lambda shape, lo, hi: torch.linspace(lo, hi, max(1, int(torch.tensor(shape).prod()) if shape else 1), **kw).reshape(shape)  # noqa: E731
    k1 = ramp(k1_shape, 2.0, 5.0)
    energy = ramp(energy_shape, 9e7, 1.1e8)
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    els = [module.Drift(t(0.4), **kw), module.Quadrupole(t(0.2), k1=k1, **kw)]
    if lattice == "cavity":
        els += [module.Cavity(t(0.5), voltage=t(5e6), phase=t(10.0), frequency=t(1.3e9), **kw)]
    els += [module.Drift(t(0.3), **kw)]
    if lattice == "cavity":
        els += [module.BPM(is_active=True, name="bpm", **kw)]
    seg = module.Segment(els)
    n = 16
    base = torch.stack([torch.linspace(-1, 1, n, **kw) * s for s in (2e-4, 3e-5, -2e-4, 2e-5, 1e-4, 1e-3)] + [torch.ones(n, **kw)], dim=-1)
    scale = ramp(beam_shape, 1.0, 1.5)
    if kind == "particle":
        x = base * scale.reshape(*beam_shape, 1, 1)
        x[..., 6] = 1.0
        beam = module.ParticleBeam(x, energy, **kw)
    else:
        mu = torch.cat([base.mean(dim=0)[:6] + 1e-4, torch.ones(1, **kw)]) * scale.reshape(*beam_shape, 1)
        mu[..., 6] = 1.0
        cov = torch.zeros(7, 7, **kw)
        cov[:6, :6] = torch.cov(base[:, :6].T)
        cov = cov * scale.reshape(*beam_shape, 1, 1).square()
        beam = module.ParameterBeam(mu, cov, energy, **kw)
    return seg, beam


def describe(out, kind, seg):
    d = {"energy": list(out.energy.shape), "s": list(out.s.shape)}
    if kind == "particle":
        d["particles"] = list(out.particles.shape)
        d["survival"] = list(out.survival_probabilities.shape)
        d["sum"] = float(out.particles[..., :6].sum())
        d["abs"] = float(out.particles[..., :6].abs().sum())
    else:
        d["mu"], d["cov"] = list(out.mu.shape), list(out.cov.shape)
        d["total_charge"] = list(out.total_charge.shape)
        d["sum"] = float(out.mu[..., :6].sum())
        d["abs"] = float(out.cov.abs().sum())
    d["energy_sum"] = float(out.energy.sum())
    if hasattr(seg, "bpm"):
        d["bpm"] = list(seg.bpm.reading.shape)
    return d

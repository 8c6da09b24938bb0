"""The autograd sweep (tests/golden/grad_flags.json): every tensor setting of every element kind made trainable in turn (and the
beam energy, and the incoming coordinates), a small beam tracked, and a loss that reaches the coordinates, the energy and the
path length — which outputs carry a graph, and d loss / d setting. The same code runs against the reference and this engine."""
import torch

SETTINGS = {
    "Drift": {"length": 0.5},
    "Quadrupole": {"length": 0.2, "k1": 3.0, "tilt": 0.1, "misalignment": [1e-4, -2e-4]},
    "Dipole": {"length": 0.5, "angle": 0.1, "k1": 0.5, "dipole_e1": 0.02, "dipole_e2": 0.03, "tilt": 0.05, "gap": 0.03, "fringe_integral": 0.4},
    "HorizontalCorrector": {"length": 0.1, "angle": 1e-4},
    "VerticalCorrector": {"length": 0.1, "angle": 1e-4},
    "Solenoid": {"length": 0.2, "k": 1.0},
    "Cavity": {"length": 0.5, "voltage": 5e6, "phase": 10.0, "frequency": 1.3e9},
    "Sextupole": {"length": 0.2, "k2": 5.0},
    "Undulator": {"length": 0.3},
    "Aperture": {"x_max": 1e-3, "y_max": 2e-3},
    "SpaceChargeKick": {"effect_length": 0.2, "grid_extent_x": 3.0, "grid_extent_y": 2.7, "grid_extent_tau": 3.4},
    "TransverseDeflectingCavity": {"length": 0.3, "voltage": 1e6, "phase": 20.0, "frequency": 2.9e9},
}
EXTRA = {"Aperture": {"is_active": True}, "SpaceChargeKick": {"grid_shape": (16, 16, 16)}, "Sextupole": {"tracking_method": "linear"}}
# a grid outside the pruned power-of-two transforms: gradients through the dense hipFFT plans (ScPoissonDense)
SETTINGS["SpaceChargeKick@odd_grid"] = dict(SETTINGS["SpaceChargeKick"])
EXTRA["SpaceChargeKick@odd_grid"] = {"grid_shape": (12, 10, 14)}
# the non-linear tracking methods: "Kind@method" builds Kind with that tracking_method
for _kind, _methods in (("Drift", ("second_order", "drift_kick_drift")), ("Quadrupole", ("second_order", "drift_kick_drift")),
                        ("Dipole", ("second_order", "drift_kick_drift")), ("Sextupole", ("second_order",))):
    for _m in _methods:
        SETTINGS[f"{_kind}@{_m}"] = {k: v for k, v in SETTINGS[_kind].items() if not (_kind == "Dipole" and _m == "drift_kick_drift" and k == "k1")}
        EXTRA[f"{_kind}@{_m}"] = {"tracking_method": _m}


def cases():
    for kind, settings in SETTINGS.items():
        for name in settings:
            yield kind, name
        yield kind, "beam.energy"
        yield kind, "beam.coordinates"


def run(module, kind, trainable, beam_kind, dev):
    kw = {"dtype": torch.float64, "device": dev}
    args, leaf = {}, None
    for name, v in SETTINGS[kind].items():
        tns = torch.tensor(v, **kw)
        if name == trainable:
            tns = leaf = torch.nn.Parameter(tns)
        args[name] = tns
    el = getattr(module, kind.split("@")[0])(**args, **EXTRA.get(kind, {}), **kw)
    n = 48
    # deterministic, but not on a regular lattice: evenly spaced coordinates put particles exactly on the nodes of a grid whose
    # extent is a multiple of the beam size, where the trilinear weights have a kink and the side taken depends on the last bit
    # of sigma (seen with linspace coordinates: d loss / d x differed from the reference by 1e-3 for that reason alone)
    i = torch.arange(n, **kw)
    x = torch.stack([torch.sin(a * i + b) * s for a, b, s in ((1.3, 0.2, 2e-4), (0.7, 1.1, 3e-5), (2.1, 0.5, -2e-4), (0.9, 2.3, 2e-5),
                                                              (1.7, 0.9, 1e-4), (0.4, 1.9, 1e-3))] + [torch.ones(n, **kw)], dim=-1)
    energy = torch.tensor(1e8, **kw)
    if trainable == "beam.energy":
        energy = leaf = torch.nn.Parameter(energy)
    if beam_kind == "particle":
        if trainable == "beam.coordinates":
            x = leaf = torch.nn.Parameter(x)
        beam = module.ParticleBeam(x, energy, particle_charges=torch.full((n,), 1e-13, **kw), **kw)
    else:
        cov = torch.zeros(7, 7, **kw)
        cov[:6, :6] = torch.cov(x[:, :6].T)
        mu = torch.cat([x.mean(dim=0)[:6] + 1e-4, torch.ones(1, **kw)])
        if trainable == "beam.coordinates":
            mu = leaf = torch.nn.Parameter(mu)
        beam = module.ParameterBeam(mu, cov, energy, **kw)
    out = el.track(beam)
    coords = out.particles if beam_kind == "particle" else out.mu
    d = {"coords": bool(coords.requires_grad), "energy": bool(out.energy.requires_grad), "s": bool(out.s.requires_grad)}
    if beam_kind == "particle":
        d["survival"] = bool(out.survival_probabilities.requires_grad)
        loss = (coords[..., :6] * torch.arange(1, 7, **kw)).sum() * 1e3 + out.energy.sum() * 1e-8 + out.s.sum() + out.survival_probabilities.sum() * 1e-3
    else:
        d["cov"] = bool(out.cov.requires_grad)
        loss = (coords[..., :6] * torch.arange(1, 7, **kw)).sum() * 1e3 + out.cov.sum() * 1e6 + out.energy.sum() * 1e-8 + out.s.sum()
    d["loss"] = float(loss)
    if loss.requires_grad:
        (g,) = torch.autograd.grad(loss, leaf, allow_unused=True)
        d["grad"] = None if g is None else [float(v) for v in g.reshape(-1)[:12]]
    else:
        d["grad"] = "no graph"
    return d

"""Solenoid / Undulator / Sextupole(linear). CPU: oracle vs reference maps; GPU: HIP maps
and the reference's consistency goldens through cheetah_amd."""
import numpy as np
import pytest
import torch


def _cases(g):
    for i in range(int(g["n_cases"])):
        yield str(g[f"m{i}_kind"]), g[f"m{i}_params"], float(g[f"m{i}_energy"]), g[f"m{i}_R"]


def _err(R, Rref):
    denom = np.maximum(np.abs(Rref), 1e-3 * np.max(np.abs(Rref)))
    return np.max(np.abs(R - Rref) / denom)


def test_oracle_maps_match_reference(golden, oracle):
    g = golden("misc_elements.npz")
    n = 0
    for kind, p, E, Rref in _cases(g):
        assert _err(oracle.build_rmatrix(kind, p, E)[0], Rref) < 1e-12, (kind, p)
        n += 1
    assert n == 18


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_hip_maps_match_reference(golden, tag):
    from cheetah_amd import _ops

    g = golden("misc_elements.npz")
    dt = torch.float64 if tag == "f64" else torch.float32
    for kind, p, E, Rref in _cases(g):
        R = _ops.build_rmatrix(_ops.KIND[kind], torch.tensor(p[None], dtype=dt, device="cuda"),
                               torch.tensor([E], dtype=dt, device="cuda"), 510998.95069, -1.0, 1).cpu().numpy()[0]
        assert _err(R, Rref) < (1e-12 if tag == "f64" else 5e-5), (kind, p)


@pytest.mark.gpu
def test_reference_consistency_goldens_misc(golden):
    import cheetah_amd as ca

    g, c = golden("misc_elements.npz"), golden("consistency.npz")
    f64 = torch.float64
    t = lambda v: torch.tensor(v, dtype=torch.float32).to(f64).cuda()  # noqa: E731
    kw = {"dtype": f64, "device": "cuda"}
    elements = {
        "Solenoid_ParticleBeam_default": ca.Solenoid(length=t(1.0), k=t([1.0, -2.0]), misalignment=t([0.01, -0.02])),
        "Undulator_ParticleBeam_default": ca.Undulator(length=t(1.0), period=t(0.1), kx=t(1.3), **kw),
        "Sextupole_ParticleBeam_linear": ca.Sextupole(length=t(1.0), k2=t([1.0, -2.0]), tilt=t(0.42), misalignment=t([0.01, -0.02]), tracking_method="linear"),
    }
    keep = int(g["keep"])
    for name, el in elements.items():
        beam = ca.ParticleBeam(torch.tensor(c["incoming_particles_f32"], device="cuda").to(f64),
                               torch.tensor(c["incoming_energy"], device="cuda").reshape(()),
                               species=ca.Species("electron", **kw))
        out = el.track(beam).particles.cpu().numpy()[..., :keep, :]
        exp = g[f"{name}__particles"]
        assert out.shape == exp.shape, name
        assert np.allclose(out, exp, rtol=1e-5, atol=1e-8), name
        # (the Undulator pickle predates a 1e-12-level change of the reference itself: 1e-11 there)
        assert np.max(np.abs(out - exp)) / np.max(np.abs(exp)) < (1e-11 if name.startswith("Undulator") else 1e-12), name

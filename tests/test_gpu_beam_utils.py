"""Derived-beam helpers of ParticleBeam against the reference (tests/golden/beam_utils.npz): make_linspaced,
linspaced, transformed_to, as_parameter_beam, randomly_subsampled. The moments they need come from chx_moments."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_derived_beams_match_reference(golden):
    import cheetah_amd as ca

    g = golden("beam_utils.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    beam = ca.ParticleBeam(t(g["in"]), t(g["energy"]), particle_charges=t(g["charges"]), species=ca.Species("electron", **kw))
    tr = beam.transformed_to(mu_x=t(1e-3), sigma_y=t([1e-4, 2e-4]), sigma_p=t(5e-4), total_charge=t(2e-12), energy=t(2e8))
    assert tuple(tr.particles.shape) == g["transformed"].shape
    assert np.allclose(tr.particles.cpu().numpy(), g["transformed"], rtol=1e-10, atol=1e-16)
    assert np.allclose(tr.particle_charges.cpu().numpy(), g["transformed_charges"], rtol=1e-14)
    assert float(tr.energy) == float(g["transformed_energy"])
    assert float(tr.mu_x.max()) == pytest.approx(1e-3, rel=1e-10) and np.allclose(tr.sigma_y.cpu().numpy(), [1e-4, 2e-4], rtol=1e-10)
    lin = ca.ParticleBeam.make_linspaced(num_particles=17, mu_x=t([1e-3, -1e-3]), sigma_px=t(3e-6), energy=t(1.5e8),
                                         total_charge=t(1e-10), **kw)
    # the reference builds each ramp with torch.linspace in the DEFAULT dtype (utils/elementwise_linspace.py:25), i.e.
    # its fp64 beams carry fp32-rounded coordinates; ours are exact, so the comparison is at fp32 resolution
    assert np.allclose(lin.particles.cpu().numpy(), g["linspaced"], rtol=2e-7, atol=1e-20)
    assert np.allclose(lin.particle_charges.cpu().numpy(), g["linspaced_charges"], rtol=1e-14)
    lin2 = beam.linspaced(33)
    # (the reference's own `linspaced` raises TypeError; check the documented meaning instead)
    assert lin2.num_particles == 33 and float(lin2.total_charge) == pytest.approx(float(beam.total_charge), rel=1e-12)
    assert float(lin2.particles[:, 0].min()) == pytest.approx(float(beam.mu_x - beam.sigma_x), rel=1e-9)
    assert float(lin2.particles[:, 0].max()) == pytest.approx(float(beam.mu_x + beam.sigma_x), rel=1e-9)
    assert float(lin2.mu_p) == pytest.approx(float(beam.mu_p), rel=1e-9)
    pb = beam.as_parameter_beam()
    assert np.allclose(pb.mu.cpu().numpy(), g["pb_mu"], rtol=1e-10, atol=1e-18)
    scale = np.sqrt(np.outer(np.diag(g["pb_cov"])[:6], np.diag(g["pb_cov"])[:6]))
    assert np.all(np.abs(pb.cov.cpu().numpy()[:6, :6] - g["pb_cov"][:6, :6]) <= 1e-10 * scale)
    assert np.all(pb.cov.cpu().numpy()[6] == 0) and float(pb.mu[6]) == 1.0
    assert float(pb.total_charge) == pytest.approx(float(g["pb_total_charge"]), rel=1e-13)
    gen = torch.Generator(device="cuda").manual_seed(3)
    sub = beam.randomly_subsampled(100, random_state=gen)
    assert sub.num_particles == 100 and float(sub.total_charge) == pytest.approx(float(beam.total_charge), rel=1e-12)
    rows = {tuple(r) for r in beam.particles.cpu().numpy().round(15).tolist()}
    assert all(tuple(r) in rows for r in sub.particles.cpu().numpy().round(15).tolist())


def test_clone_many_copies_every_array_in_one_launch():
    """_ops.clone_many (chx_copy_arrays): sizes from one scalar to a particle array, unaligned views, dtypes mixed, an empty
    tensor, a non-contiguous one and one that carries gradients (those two fall back to `clone()`)."""
    import torch

    from cheetah_amd import _ops

    torch.manual_seed(0)
    big = torch.randn(100_003, 7, device="cuda")
    base = torch.randn(1000, device="cuda", dtype=torch.float64)
    g = torch.randn(5, device="cuda", requires_grad=True)
    tensors = [big, torch.tensor(3.5, device="cuda"), base[1:], torch.empty(0, device="cuda"), big[:, ::2], g,
               torch.arange(17, device="cuda", dtype=torch.int32), base[3:4], torch.randn(33, device="cuda", dtype=torch.float16),
               torch.randn(9, device="cuda")]
    copies = _ops.clone_many(tensors)
    assert len(copies) == len(tensors)
    for t, c in zip(tensors, copies):
        assert c.shape == t.shape and c.dtype == t.dtype and torch.equal(c, t)
        assert t.numel() == 0 or c.data_ptr() != t.data_ptr()
    assert copies[5].requires_grad and copies[5].grad_fn is not None        # still connected to the graph
    with torch.no_grad():
        assert not _ops.clone_many([g])[0].requires_grad
    big.add_(1.0)
    assert not torch.equal(copies[0], big)


@pytest.mark.gpu
def test_every_beam_property_on_drawn_beams_vs_reference(golden):
    """beam_properties_random.npz (tests/golden/generate_golden_random_beam_properties.py): all derived properties of the
    reference's Beam / ParticleBeam / ParameterBeam classes on eight drawn beams — correlated coordinates, unequal and negative
    charges, survival probabilities with exact zeros, four species, gamma 1.3 … 1e4, vector shapes on the particles and / or the
    energy — value and shape, in float64. Properties the reference itself cannot evaluate for a shape (`energies` of shared
    particles with a vectorised energy) are not in the fixture and not asserted."""
    import cheetah_amd as ca

    g = golden("beam_properties_random.npz")
    kw = {"dtype": torch.float64, "device": "cuda"}
    base, moments, ponly = [str(v) for v in g["base"]], [str(v) for v in g["moments"]], [str(v) for v in g["particle_only"]]

    def check(obj, name, ref, what):
        got = getattr(obj, name)
        got = got.cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
        assert got.shape == ref.shape, (what, name, got.shape, ref.shape)
        # second moments of 200-700 weighted particles: differences of products, 1e-9 of the reference's own scale is the
        # cancellation in its two-pass formula; the dispersions divide two such numbers
        scale = np.maximum(np.abs(ref), 1e-12 * max(np.abs(ref).max(), 1e-300))
        assert np.all(np.abs(got - ref) <= 2e-9 * scale + 1e-300), (what, name, got, ref)

    for i in range(int(g["n_beams"])):
        name = str(g[f"species_{i}"])
        sp = (ca.Species("ion", num_elementary_charges=torch.tensor(6.0, **kw), mass_eV=torch.tensor(1.1178e10, **kw)) if name == "ion"
              else ca.Species(name, **kw))
        beam = ca.ParticleBeam(torch.tensor(g[f"particles_{i}"], **kw), torch.tensor(g[f"energy_{i}"], **kw),
                               particle_charges=torch.tensor(g[f"charges_{i}"], **kw),
                               survival_probabilities=torch.tensor(g[f"survival_{i}"], **kw), species=sp)
        for prop in base + moments + ponly:
            if f"pb_{prop}_{i}" in g.files:
                check(beam, prop, g[f"pb_{prop}_{i}"], f"ParticleBeam {i}")
        assert beam.num_particles == int(g[f"pb_num_particles_{i}"])
        conv = beam.as_parameter_beam()
        mu_ref, cov_ref = g[f"mu_{i}"], g[f"cov_{i}"]
        assert conv.mu.shape == mu_ref.shape and conv.cov.shape == cov_ref.shape
        sig = np.sqrt(np.abs(np.diagonal(cov_ref, axis1=-2, axis2=-1)))[..., :6]
        assert np.abs(conv.mu.cpu().numpy() - mu_ref).max() <= 1e-12 * np.abs(mu_ref).max()
        cov = conv.cov.cpu().numpy()
        assert (np.abs(cov - cov_ref)[..., :6, :6] / (sig[..., :, None] * sig[..., None, :])).max() < 1e-9
        # the affine coordinate has no spread: exactly 0 here, rounding noise of the reference's weighted two-pass formula there
        assert np.abs(cov[..., 6, :]).max() == 0.0 and np.abs(cov[..., :, 6]).max() == 0.0 and np.abs(cov_ref[..., 6, :]).max() < 1e-18
        par = ca.ParameterBeam(torch.tensor(mu_ref, **kw), torch.tensor(cov_ref, **kw), torch.tensor(g[f"energy_{i}"], **kw),
                               total_charge=beam.total_charge, species=sp)
        for prop in base + moments:
            if f"par_{prop}_{i}" in g.files:
                check(par, prop, g[f"par_{prop}_{i}"], f"ParameterBeam {i}")

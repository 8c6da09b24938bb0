#!/usr/bin/env python3
"""Golden vectors for the non-linear tracking methods: tests/golden/dkd.npz and
tests/golden/second_order.npz.

Run in the build container only (imports /root/reference read-only through generate_golden.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_nonlinear.py

Contents are DATA only:
  * the Bmad-X results the reference's tests compare against (tests/resources/bmadx/*.pt, used by
    tests/test_drift.py:41-70, test_quadrupole.py:172-207, test_dipole.py:103-150,
    test_transverse_deflecting_cavity.py:10-41), first KEEP particles, as arrays;
  * the Bmad/Tao single-particle results for six species (tests/resources/bmad/x_tao_*.pt, test_compare_bmad.py);
  * the reference's consistency goldens for drift_kick_drift / second_order / TDC (test_elements.py:356-431);
  * outputs of the reference itself for extra cases (fp64 and fp32, vectorised, other species, fringe variants,
    second-order tensors).
"""

import copy
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from generate_golden import REF, cheetah, load_incoming, np, npy, save, species_meta, torch  # noqa: E402

from cheetah.utils.bmadx import bmad_to_cheetah_z_pz  # noqa: E402
from scipy.constants import physical_constants  # noqa: E402

KEEP = 2048
F64 = {"dtype": torch.float64}


def t64(v):
    return torch.tensor(v, **F64)


def bmadx_elements(dtype):
    """The four elements of the reference's Bmad-X comparison tests, as (name, element, kind, params, steps, fringe)."""
    kw = {"dtype": dtype}
    angle = torch.tensor(20 * torch.pi / 180, **kw)
    e1 = angle / 2
    e2 = angle - e1
    drift = cheetah.Drift(length=torch.tensor(1.0), tracking_method="drift_kick_drift", **kw)
    quad = cheetah.Quadrupole(length=torch.tensor(1.0, **kw), k1=torch.tensor(10.0, **kw),
                              misalignment=torch.tensor([0.01, -0.02], **kw), tilt=torch.tensor(0.5, **kw), num_steps=10,
                              tracking_method="drift_kick_drift", **kw)
    dip = cheetah.Dipole(length=torch.tensor(0.5), angle=angle, dipole_e1=e1, dipole_e2=e2, tilt=torch.tensor(0.1, **kw),
                         fringe_integral=torch.tensor(0.5), fringe_integral_exit=torch.tensor(0.5),
                         gap=torch.tensor(0.05, **kw), gap_exit=torch.tensor(0.05, **kw), fringe_at="both",
                         fringe_type="linear_edge", tracking_method="drift_kick_drift", **kw)
    tdc = cheetah.TransverseDeflectingCavity(length=torch.tensor(1.0, **kw), voltage=torch.tensor(1e7, **kw),
                                             phase=torch.tensor(0.2, **kw), frequency=torch.tensor(1e9, **kw),
                                             tracking_method="drift_kick_drift", **kw)
    return {"drift": drift, "quadrupole": quad, "dipole": dip, "tdc": tdc}


def dkd_params(el):
    """Parameter vector of the C-ABI for a reference element (include/chx.h, chx_dkd_track)."""
    f = lambda v: npy(v.to(torch.float64))  # noqa: E731
    if isinstance(el, cheetah.TransverseDeflectingCavity):
        b = torch.broadcast_tensors(el.length, el.voltage, el.phase, el.frequency, el.tilt, el.misalignment[..., 0],
                                    el.misalignment[..., 1])
        return "tdc", np.stack([f(v) for v in b], axis=-1), 1, 3
    if isinstance(el, cheetah.Dipole):
        b = torch.broadcast_tensors(el.length, el.angle, el.dipole_e1, el.dipole_e2, el.tilt, el.fringe_integral,
                                    el.fringe_integral_exit, el.gap, el.gap_exit)
        fr = {"neither": 0, "entrance": 1, "exit": 2, "both": 3}[el.fringe_at]
        return "dipole", np.stack([f(v) for v in b], axis=-1), 1, fr
    if isinstance(el, cheetah.Quadrupole):
        b = torch.broadcast_tensors(el.length, el.k1, el.tilt, el.misalignment[..., 0], el.misalignment[..., 1])
        return "quadrupole", np.stack([f(v) for v in b], axis=-1), el.num_steps, 3
    return "drift", f(el.length)[..., None], 1, 3


def gen_dkd():
    arrays = {}
    inc = torch.load(os.path.join(REF, "tests/resources/bmadx/incoming.pt"), weights_only=False)
    arrays["incoming"] = npy(inc.particles)[:KEEP]
    arrays["energy"] = npy(inc.energy)
    arrays["species"] = np.asarray(species_meta(inc.species))
    files = {"drift": "outgoing_drift.pt", "quadrupole": "outgoing_quadrupole.pt", "dipole": "outgoing_dipole.pt",
             "tdc": "outgoing_transverse_deflecting_cavity.pt"}
    els = bmadx_elements(torch.float64)
    els32 = bmadx_elements(torch.float32)
    for name, fn in files.items():
        out = torch.load(os.path.join(REF, "tests/resources/bmadx", fn), weights_only=False)
        arrays[f"bmadx_{name}"] = npy(out).reshape(-1, 7)[:KEEP]
        kind, params, steps, fringe = dkd_params(els[name])
        arrays[f"bmadx_{name}_params"] = params
        arrays[f"bmadx_{name}_steps"] = np.asarray(steps)
        arrays[f"bmadx_{name}_fringe"] = np.asarray(fringe)
        # the reference's own fp64 / fp32 results on the same particles
        o64 = els[name].track(inc)
        arrays[f"ref64_{name}"] = npy(o64.particles).reshape(-1, 7)[:KEEP]
        arrays[f"ref64_{name}_energy"] = npy(o64.energy)
        o32 = els32[name].track(copy.deepcopy(inc).to(torch.float32))  # Module.to() works in place
        arrays[f"ref32_{name}"] = npy(o32.particles).reshape(-1, 7)[:KEEP]

    # Bmad / Tao single-particle comparison (tests/test_compare_bmad.py)
    atomic_mass_eV = physical_constants["atomic mass constant energy equivalent in MeV"][0] * 1e6
    species = {
        "proton": cheetah.Species("proton", **F64), "electron": cheetah.Species("electron", **F64),
        "positron": cheetah.Species("positron", **F64), "antiproton": cheetah.Species("antiproton", **F64),
        "deuteron": cheetah.Species("deuteron", **F64),
        "#12C+3": cheetah.Species("#12C+3", num_elementary_charges=t64(3.0), mass_eV=t64(12.0) * atomic_mass_eV, **F64),
    }
    tao_els = {
        "Drift": cheetah.Drift(length=torch.tensor(1.0), tracking_method="drift_kick_drift", **F64),
        "Dipole": cheetah.Dipole(length=torch.tensor(0.5), angle=torch.tensor(0.2), dipole_e1=torch.tensor(0.1),
                                 dipole_e2=torch.tensor(0.1), tilt=torch.tensor(0.1), fringe_integral=torch.tensor(0.5),
                                 fringe_integral_exit=torch.tensor(0.5), gap=torch.tensor(0.06), gap_exit=torch.tensor(0.06),
                                 fringe_at="both", fringe_type="linear_edge", tracking_method="drift_kick_drift", **F64),
        "Quadrupole": cheetah.Quadrupole(length=torch.tensor(0.5), k1=torch.tensor(1.0), tracking_method="drift_kick_drift",
                                         **F64),
    }
    coords = t64([1e-3, 2e-3, -3e-3, -1e-3, 2e-3, -1e-3])
    p0c = t64(5.0e7)
    arrays["tao_coords"] = npy(coords)
    arrays["tao_p0c"] = npy(p0c)
    arrays["tao_species_names"] = np.asarray(["proton", "electron", "positron", "antiproton", "deuteron", "12C+3"])
    for ename, el in tao_els.items():
        _, params, steps, fringe = dkd_params(el)
        arrays[f"tao_{ename}_params"] = params
        arrays[f"tao_{ename}_fringe"] = np.asarray(fringe)
    for sname, sp in species.items():
        tau, delta, ref_energy = bmad_to_cheetah_z_pz(coords[4], coords[5], p0c, sp.mass_eV)
        key = sname.replace("#", "")
        arrays[f"tao_{key}_species"] = np.asarray(species_meta(sp))
        arrays[f"tao_{key}_tau_delta_energy"] = np.asarray([float(tau), float(delta), float(ref_energy)])
        for ename in tao_els:
            arrays[f"tao_{key}_{ename}"] = npy(torch.load(os.path.join(REF, "tests/resources/bmad", f"x_tao_{sname}_{ename}.pt")))

    # extra cases run through the reference (fp64): vectorised elements / energies, fringe variants, zero angle,
    # negative k1, protons at low energy (low_energy_z_correction's other branch), misaligned + tilted TDC
    sub = cheetah.ParticleBeam(particles=inc.particles[:512], energy=inc.energy, species=inc.species, **F64)
    prot = cheetah.Species("proton", **F64)
    sub_p = cheetah.ParticleBeam(particles=inc.particles[:512] * t64([1, 1, 1, 1, 1, 0.01, 1]), energy=t64(9.6e8), species=prot, **F64)
    sub_e = cheetah.ParticleBeam(particles=inc.particles[:512], energy=t64([[5e7], [2e8]]), species=inc.species, **F64)
    sub_e1 = cheetah.ParticleBeam(particles=inc.particles[:512], energy=t64([5e7, 2e8]), species=inc.species, **F64)
    dk = {"tracking_method": "drift_kick_drift", **F64}
    extra = [
        ("drift_vec", cheetah.Drift(length=t64([1.0, -1.0, 0.3]), **dk), sub),
        ("drift_proton", cheetah.Drift(length=t64(2.0), **dk), sub_p),
        ("quad_vec", cheetah.Quadrupole(length=t64(1.0), k1=t64([1.0, -2.0, 0.0]), tilt=t64(0.42),
                                        misalignment=t64([0.01, -0.02]), **dk), sub),
        ("quad_steps5_proton", cheetah.Quadrupole(length=t64(0.4), k1=t64(-3.0), num_steps=5, **dk), sub_p),
        ("quad_energy_vec", cheetah.Quadrupole(length=t64(0.2), k1=t64([4.2, -4.2]), num_steps=3, **dk), sub_e),
        ("dipole_vec", cheetah.Dipole(length=t64(1.0), angle=t64([1.0, -2.0]), tilt=t64(0.42), **dk), sub),
        ("dipole_zero", cheetah.Dipole(length=t64(1.0), angle=t64(0.0), **dk), sub),
        ("dipole_entrance", cheetah.Dipole(length=t64(0.7), angle=t64(0.3), dipole_e1=t64(0.1), dipole_e2=t64(0.2),
                                           fringe_integral=t64(0.4), fringe_integral_exit=t64(0.6), gap=t64(0.03),
                                           gap_exit=t64(0.05), fringe_at="entrance", **dk), sub),
        ("dipole_exit_proton", cheetah.Dipole(length=t64(0.7), angle=t64(-0.3), dipole_e1=t64(0.1), dipole_e2=t64(0.2),
                                              fringe_integral=t64(0.4), fringe_integral_exit=t64(0.6), gap=t64(0.03),
                                              gap_exit=t64(0.05), fringe_at="exit", **dk), sub_p),
        ("dipole_neither", cheetah.Dipole(length=t64(0.7), angle=t64(0.3), dipole_e1=t64(0.1), fringe_at="neither", **dk), sub),
        ("rbend", cheetah.RBend(length=t64(1.0), angle=t64([1.0, -2.0]), tilt=t64(0.42), **dk), sub),
        ("tdc_vec", cheetah.TransverseDeflectingCavity(length=t64(1.0), voltage=t64([[1e7], [2e7], [0.0]]), phase=t64(0.4),
                                                       frequency=t64(1e9), **dk), sub_e1),
        ("tdc_misaligned", cheetah.TransverseDeflectingCavity(length=t64(0.8), voltage=t64(5e6), phase=t64(-0.1),
                                                              frequency=t64(2.856e9), tilt=t64(0.3),
                                                              misalignment=t64([1e-3, -2e-3]), **dk), sub_p),
    ]
    names = []
    for name, el, beam in extra:
        out = el.track(beam)
        kind, params, steps, fringe = dkd_params(el)
        arrays[f"x_{name}_kind"] = np.asarray(kind)
        arrays[f"x_{name}_params"] = params
        arrays[f"x_{name}_steps"] = np.asarray(steps)
        arrays[f"x_{name}_fringe"] = np.asarray(fringe)
        arrays[f"x_{name}_in"] = npy(beam.particles)
        arrays[f"x_{name}_energy"] = npy(beam.energy)
        arrays[f"x_{name}_species"] = np.asarray(species_meta(beam.species))
        arrays[f"x_{name}_out"] = npy(out.particles)
        arrays[f"x_{name}_energy_out"] = npy(out.energy)
        names.append(name)
    arrays["extra_names"] = np.asarray(names)

    # consistency goldens (incoming = tests/golden/consistency.npz's beam)
    for cname in ("Drift_ParticleBeam_drift_kick_drift", "Quadrupole_ParticleBeam_drift_kick_drift",
                  "Dipole_ParticleBeam_drift_kick_drift", "RBend_ParticleBeam_drift_kick_drift",
                  "TransverseDeflectingCavity_ParticleBeam_active", "TransverseDeflectingCavity_ParticleBeam_inactive"):
        with open(os.path.join(REF, "tests/resources/consistency_expected_outgoing", cname + ".pkl"), "rb") as f:
            exp = pickle.load(f)
        arrays[f"{cname}__particles"] = npy(exp.particles)[..., :512, :]
        arrays[f"{cname}__energy"] = npy(exp.energy)
    save("dkd.npz", **arrays)


def t_params(el):
    f = lambda v: npy(v.to(torch.float64))  # noqa: E731
    if isinstance(el, cheetah.Dipole):
        b = torch.broadcast_tensors(el.length, el.angle, el.k1, el.dipole_e1, el.dipole_e2, el.tilt, el.fringe_integral,
                                    el.fringe_integral_exit, el.gap)
        return "dipole", np.stack([f(v) for v in b], axis=-1)
    if isinstance(el, cheetah.Quadrupole):
        b = torch.broadcast_tensors(el.length, el.k1, el.tilt, el.misalignment[..., 0], el.misalignment[..., 1])
        return "quadrupole", np.stack([f(v) for v in b], axis=-1)
    if isinstance(el, cheetah.Sextupole):
        b = torch.broadcast_tensors(el.length, el.k2, el.tilt, el.misalignment[..., 0], el.misalignment[..., 1])
        return "sextupole", np.stack([f(v) for v in b], axis=-1)
    return "drift", f(el.length)[..., None]


def gen_second_order():
    arrays = {}
    incoming = load_incoming().to(torch.float64)
    elec = incoming.species
    prot = cheetah.Species("proton", **F64)
    so = {"tracking_method": "second_order", **F64}
    cases = [
        ("drift", cheetah.Drift(length=t64([1.0, -1.0, 0.25]), **so), t64(1e8), elec),
        ("drift_lowE", cheetah.Drift(length=t64(0.7), **so), t64([6e6, 2e7]), elec),
        ("quad", cheetah.Quadrupole(length=t64(1.0), k1=t64([1.0, -2.0, 0.0, 30.0]), tilt=t64(0.42),
                                    misalignment=t64([0.01, -0.02]), **so), t64(1.0732e8), elec),
        ("quad_small", cheetah.Quadrupole(length=t64(0.2), k1=t64([4.2, -4.2, 1e-3, -1e-6]), **so), t64(1e8), elec),
        ("quad_proton", cheetah.Quadrupole(length=t64(0.5), k1=t64(1.5), tilt=t64(-0.2), **so), t64(1.5e9), prot),
        ("dipole", cheetah.Dipole(length=t64(1.0), angle=t64([1.0, -2.0, 1e-3, 0.0]), tilt=t64(0.42), **so), t64(1e8), elec),
        ("dipole_full", cheetah.Dipole(length=t64(0.5), angle=t64(0.2), k1=t64([0.0, 1.3, -0.8]), dipole_e1=t64(0.1),
                                       dipole_e2=t64(0.05), tilt=t64(0.1), fringe_integral=t64(0.5),
                                       fringe_integral_exit=t64(0.4), gap=t64(0.06), **so), t64(6e6), elec),
        ("rbend", cheetah.RBend(length=t64(1.0), angle=t64([1.0, -2.0]), tilt=t64(0.42), **so), t64(1e8), elec),
        ("sextupole", cheetah.Sextupole(length=t64(1.0), k2=t64([1.0, -2.0, 0.0]), tilt=t64(0.42),
                                        misalignment=t64([0.01, -0.02]), **so), t64(1e8), elec),
    ]
    names = []
    x = incoming.particles[:512]
    arrays["incoming"] = npy(x)
    arrays["incoming_f32"] = npy(x.to(torch.float32))
    for name, el, energy, sp in cases:
        kind, params = t_params(el)
        T = el.second_order_transfer_map(energy, sp)
        beam = cheetah.ParticleBeam(particles=x, energy=energy, species=sp, **F64)
        out = el.track(beam)
        arrays[f"{name}_kind"] = np.asarray(kind)
        arrays[f"{name}_params"] = params
        arrays[f"{name}_energy"] = npy(energy)
        arrays[f"{name}_species"] = np.asarray(species_meta(sp))
        arrays[f"{name}_T"] = npy(T)
        arrays[f"{name}_out"] = npy(out.particles)
        # the same in fp32 (element, beam and tensors in fp32)
        el32 = el.clone().to(torch.float32) if hasattr(el, "clone") else el.to(torch.float32)
        beam32 = cheetah.ParticleBeam(particles=x.to(torch.float32), energy=energy.to(torch.float32),
                                      species=cheetah.Species(sp.name, dtype=torch.float32), dtype=torch.float32)
        arrays[f"{name}_out_f32"] = npy(el32.track(beam32).particles)
        names.append(name)
    arrays["names"] = np.asarray(names)
    for cname in ("Drift_ParticleBeam_second_order", "Quadrupole_ParticleBeam_second_order",
                  "Dipole_ParticleBeam_second_order", "RBend_ParticleBeam_second_order",
                  "Sextupole_ParticleBeam_second_order"):
        with open(os.path.join(REF, "tests/resources/consistency_expected_outgoing", cname + ".pkl"), "rb") as f:
            exp = pickle.load(f)
        arrays[f"{cname}__particles"] = npy(exp.particles)[..., :512, :]
    save("second_order.npz", **arrays)


def gen_aperture():
    """Row f3: Aperture survival masks (aperture.py:90-135) incl. particles exactly on / one ulp off the boundary,
    both shapes, fp32 and fp64, vectorised limits, pre-existing survival probabilities; plus the lattice utilities
    (split / merge / Superimposed / inactive_elements_as_drifts) as tracked outputs of the edited lattices."""
    arrays = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        g = torch.Generator().manual_seed(7)
        N = 4096
        x = torch.randn(N, 7, generator=g, dtype=torch.float64) * torch.tensor([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0], dtype=torch.float64)
        x = x.to(dt)
        x[:, 6] = 1.0
        xm, ym = torch.tensor(2.5e-4, dtype=dt), torch.tensor(1.5e-4, dtype=dt)
        # boundary cases: exactly on the rectangle edges, one ulp inside / outside, on the ellipse
        eps = torch.finfo(dt).eps
        edge = [xm, -xm, xm * (1 - eps), xm * (1 + eps), -xm * (1 - eps), -xm * (1 + eps)]
        for i, v in enumerate(edge):
            x[i, 0], x[i, 2] = v, 0.0
            x[8 + i, 0], x[8 + i, 2] = 0.0, v / xm * ym
        th = torch.linspace(0, 6.283, 64, dtype=dt)
        x[32:96, 0], x[32:96, 2] = xm * th.cos(), ym * th.sin()
        surv = torch.rand(N, generator=g, dtype=torch.float64).to(dt)
        surv[::7] = 0.0
        arrays[f"x_{tag}"], arrays[f"surv_{tag}"] = npy(x), npy(surv)
        sp = cheetah.Species("electron", dtype=dt)
        for shape in ("rectangular", "elliptical"):
            beam = cheetah.ParticleBeam(x, torch.tensor(1e8, dtype=dt), survival_probabilities=surv, species=sp, dtype=dt)
            out = cheetah.Aperture(x_max=xm, y_max=ym, shape=shape, dtype=dt).track(beam)
            arrays[f"{shape}_{tag}"] = npy(out.survival_probabilities)
            xv = torch.tensor([2.5e-4, 1e-4, float("inf")], dtype=dt)
            yv = torch.tensor([[1.5e-4], [3e-4]], dtype=dt)
            outv = cheetah.Aperture(x_max=xv, y_max=yv, shape=shape, dtype=dt).track(beam)
            arrays[f"{shape}_vec_{tag}"] = npy(outv.survival_probabilities)
        arrays[f"limits_{tag}"] = np.asarray([float(xm), float(ym)])
    # lattice utilities, fp64, on the consistency beam
    inc = load_incoming().to(torch.float64)
    # a fresh fp64 species: the pickled beam's species carries an fp32-rounded mass that the reference silently
    # replaces by the full-precision one in the first `species.clone()` (element.py:190)
    beam = cheetah.ParticleBeam(inc.particles[:512], inc.energy, species=cheetah.Species("electron", **F64), **F64)
    seg = cheetah.Segment([
        cheetah.Drift(t64(0.3), name="d1", **F64), cheetah.Drift(t64(0.2), name="d2", **F64),
        cheetah.Quadrupole(t64(0.1), k1=t64(4.2), name="q1a", **F64), cheetah.Quadrupole(t64(0.15), k1=t64(-2.0), name="q1b", **F64),
        cheetah.Marker(name="m", **F64),
        cheetah.HorizontalCorrector(t64(0.05), angle=t64(0.0), name="hc", **F64),
        cheetah.Solenoid(t64(0.2), k=t64(0.8), name="s1", **F64), cheetah.Solenoid(t64(0.1), k=t64(0.4), name="s2", **F64),
        cheetah.Superimposed(cheetah.Quadrupole(t64(0.4), k1=t64(1.1), name="qs", **F64),
                             cheetah.Aperture(x_max=t64(1e-4), y_max=t64(2e-4), name="ap", **F64), name="sup", **F64),
        cheetah.Drift(t64(0.5), name="d3", **F64),
    ])
    out = seg.track(beam)
    arrays["lat_in"], arrays["lat_energy"] = npy(beam.particles), npy(beam.energy)
    arrays["lat_species"] = np.asarray(species_meta(beam.species))  # the pickled beam carries an fp32-rounded mass
    arrays["lat_out"], arrays["lat_out_survival"] = npy(out.particles), npy(out.survival_probabilities)
    merged = seg.with_consecutive_elements_merged()
    arrays["lat_merged_names"] = np.asarray(merged.element_names)
    arrays["lat_merged_lengths"] = np.asarray([float(e.length) for e in merged.elements])
    arrays["lat_merged_out"] = npy(merged.track(beam).particles)
    arrays["lat_merged_q_k1"] = np.asarray(float(merged.elements[1].k1))
    arrays["lat_merged_s_k"] = np.asarray(float(merged.elements[4].k))
    drifts = seg.inactive_elements_as_drifts()
    arrays["lat_drifts_types"] = np.asarray([type(e).__name__ for e in drifts.elements])
    arrays["lat_drifts_out"] = npy(drifts.track(beam).particles)
    split = cheetah.Segment(seg.split(t64(0.12)))
    arrays["lat_split_names"] = np.asarray(split.element_names)
    arrays["lat_split_out"] = npy(split.track(beam).particles)
    arrays["lat_sigma_x_along"] = npy(seg.get_beam_attrs_along_segment("sigma_x", beam))
    mu_x, s = seg.get_beam_attrs_along_segment(("mu_x", "s"), beam, resolution=0.25)
    arrays["lat_mu_x_res"], arrays["lat_s_res"] = npy(mu_x), npy(s)
    sub = seg.subcell("q1a", "s1", include_end=False)
    arrays["lat_subcell_names"] = np.asarray(sub.element_names)
    arrays["lat_nomarkers_names"] = np.asarray(seg.without_inactive_markers().element_names)
    arrays["lat_nozero_names"] = np.asarray(seg.without_inactive_zero_length_elements().element_names)
    arrays["lat_reversed_names"] = np.asarray(seg.reversed().element_names)
    save("aperture_lattice.npz", **arrays)


def gen_ares():
    """Row f4: the ARES lattice (docs/examples/ARESlatticeStage3v1_9.json, 195 elements) re-exported through the
    reference's own LatticeJSON writer -> tests/golden/ares_lattice.json, and the reference's tracking result through
    it with a working point set (quadrupoles, correctors, two accelerating cavities, the EA screen active)."""
    import json

    src = os.path.join(REF, "docs/examples/ARESlatticeStage3v1_9.json")
    seg = cheetah.Segment.from_lattice_json(src, **F64)
    out_json = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ares_lattice.json")
    seg.to_lattice_json(out_json, title="ARES (re-exported)", info="docs/examples/ARESlatticeStage3v1_9.json via cheetah.Segment.to_lattice_json")
    print("wrote ares_lattice.json:", os.path.getsize(out_json) // 1024, "KiB")
    with open(out_json) as f:
        d = json.load(f)
    settings = {"AREAMQZM1": ("k1", 8.2), "AREAMQZM2": ("k1", -14.3), "AREAMQZM3": ("k1", 3.142), "AREAMCVM1": ("angle", 9e-5),
                "AREAMCHM1": ("angle", -1e-4), "ARLIRSBL1": ("voltage", 2.0e7), "ARLIRSBL2": ("voltage", 1.5e7),
                "ARLIMSOG1A": ("k", 0.3)}
    names = [n for n in settings if n in d["elements"]]
    for n in names:
        setattr(getattr(seg, n), settings[n][0], t64(settings[n][1]))
    seg.ARLIRSBL2.phase = t64(-10.0)
    seg.AREABSCR1.is_active = True
    seg.AREABSCR1.method = "cloud-in-cell"
    torch.manual_seed(11)
    beam = cheetah.ParticleBeam.from_parameters(num_particles=2000, energy=t64(6e6), sigma_p=t64(1e-3), **F64)
    out = seg.track(beam)
    arrays = {"names": np.asarray(names), "values": np.asarray([settings[n][1] for n in names]),
              "attrs": np.asarray([settings[n][0] for n in names]),
              "incoming": npy(beam.particles), "energy_in": npy(beam.energy), "charges": npy(beam.particle_charges),
              "outgoing": npy(out.particles), "energy_out": npy(out.energy), "s_out": npy(out.s),
              "length": npy(seg.length)}
    img = npy(seg.AREABSCR1.reading)
    nz = np.nonzero(img)
    arrays["screen_idx"] = np.stack(nz, -1).astype(np.int32)
    arrays["screen_val"] = img[nz]
    arrays["screen_shape"] = np.asarray(img.shape)
    save("ares_track.npz", **arrays)


def gen_beam_utils():
    """Derived-beam helpers of ParticleBeam (particle_beam.py:668-800, 1034-1262): make_linspaced, linspaced,
    transformed_to, as_parameter_beam on a fixed beam (fp64)."""
    arrays = {}
    inc = load_incoming().to(torch.float64)
    beam = cheetah.ParticleBeam(inc.particles[:1024], inc.energy, particle_charges=inc.particle_charges[:1024],
                                species=cheetah.Species("electron", **F64), **F64)
    arrays["in"], arrays["energy"], arrays["charges"] = npy(beam.particles), npy(beam.energy), npy(beam.particle_charges)
    tr = beam.transformed_to(mu_x=t64(1e-3), sigma_y=t64([1e-4, 2e-4]), sigma_p=t64(5e-4), total_charge=t64(2e-12),
                             energy=t64(2e8))
    arrays["transformed"], arrays["transformed_charges"] = npy(tr.particles), npy(tr.particle_charges)
    arrays["transformed_energy"] = npy(tr.energy)
    lin = cheetah.ParticleBeam.make_linspaced(num_particles=17, mu_x=t64([1e-3, -1e-3]), sigma_px=t64(3e-6),
                                              energy=t64(1.5e8), total_charge=t64(1e-10), **F64)
    arrays["linspaced"], arrays["linspaced_charges"] = npy(lin.particles), npy(lin.particle_charges)
    # (ParticleBeam.linspaced of the reference raises TypeError — it passes particle_charges to make_linspaced)
    pb = beam.as_parameter_beam()
    arrays["pb_mu"], arrays["pb_cov"], arrays["pb_total_charge"] = npy(pb.mu), npy(pb.cov), npy(pb.total_charge)
    save("beam_utils.npz", **arrays)


def gen_cavity_grad():
    """Gradients through Cavity.track (cavity.py:100-251) for a ParticleBeam: d loss / d (voltage, phase, frequency,
    length, incoming energy, incoming particles) with loss = sum(W * outgoing particles) + 1e-9 * outgoing energy,
    standing- and travelling-wave, accelerating and decelerating, fp64."""
    arrays = {}
    g = torch.Generator().manual_seed(5)
    N = 2000
    x = torch.randn(N, 7, generator=g, **F64) * t64([2e-4, 4e-6, 2e-4, 4e-6, 8e-6, 2e-3, 0.0])
    x[:, 6] = 1.0
    W = torch.randn(N, 7, generator=g, **F64)
    arrays["x"], arrays["W"] = npy(x), npy(W)
    cases = [("sw_acc", "standing_wave", 1.0377, 18.15975e6, 30.0, 1.3e9, 6e6),
             ("tw_acc", "traveling_wave", 4.139, 2.0e7, -12.0, 2.998e9, 1e8),
             ("sw_dec", "standing_wave", 1.0377, 5.0e6, 170.0, 1.3e9, 5e7)]
    for name, ctype, L, V, ph, f, E in cases:
        length = torch.nn.Parameter(t64(L))
        voltage, phase, freq = torch.nn.Parameter(t64(V)), torch.nn.Parameter(t64(ph)), torch.nn.Parameter(t64(f))
        energy = t64(E).requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        cav = cheetah.Cavity(length=length, voltage=voltage, phase=phase, frequency=freq, cavity_type=ctype, **F64)
        beam = cheetah.ParticleBeam(xin, energy, species=cheetah.Species("electron", **F64), **F64)
        out = cav.track(beam)
        loss = (out.particles * W).sum() + 1e-9 * out.energy
        loss.backward()
        arrays[f"{name}_params"] = np.asarray([L, V, ph, f, E])
        arrays[f"{name}_type"] = np.asarray(ctype)
        arrays[f"{name}_loss"] = npy(loss)
        arrays[f"{name}_out"] = npy(out.particles)
        arrays[f"{name}_grads"] = np.asarray([float(length.grad), float(voltage.grad), float(phase.grad), float(freq.grad),
                                              float(energy.grad)])
        arrays[f"{name}_dx"] = npy(xin.grad)
    arrays["names"] = np.asarray([c[0] for c in cases])
    save("cavity_grad.npz", **arrays)


def gen_nonlinear_grad():
    """Gradients through drift_kick_drift and second_order tracking (the reference differentiates them with torch
    autograd through utils/bmadx.py / track_methods.py:80-296): d loss / d (every element parameter, incoming energy,
    incoming particles) with loss = sum(W * outgoing particles) + 1e-9 * outgoing energy, fp64.
    NB the reference's second-order gradients are NaN wherever kx2 = k1 + hx^2 is exactly 0 (Drift, Sextupole): the
    unselected branch of `torch.where(kx2 != 0, ...)` in track_methods.py:134-143 evaluates 0/0 and poisons autograd.
    Those NaNs are stored as they come; the tests check such entries against finite differences instead."""
    arrays = {}
    g = torch.Generator().manual_seed(11)
    N = 600
    x = torch.randn(N, 7, generator=g, **F64) * t64([3e-4, 5e-5, 3e-4, 5e-5, 2e-5, 2e-3, 0.0])
    x[:, 6] = 1.0
    W = torch.randn(N, 7, generator=g, **F64)
    arrays["x"], arrays["W"] = npy(x), npy(W)
    dip = dict(length=0.5, angle=0.2, dipole_e1=0.08, dipole_e2=0.05, tilt=0.1, fringe_integral=0.5,
               fringe_integral_exit=0.4, gap=0.05)
    cases = [
        ("dkd_drift", "Drift", "drift_kick_drift", dict(length=0.8), {}, 1e8),
        ("dkd_drift_lowE", "Drift", "drift_kick_drift", dict(length=0.8), {}, 4e6),
        ("dkd_quad", "Quadrupole", "drift_kick_drift", dict(length=0.3, k1=4.2, tilt=0.3, misalignment=[1e-3, -2e-3]),
         dict(num_steps=5), 1e8),
        ("dkd_quad_defocus", "Quadrupole", "drift_kick_drift", dict(length=0.2, k1=-7.0), dict(num_steps=1), 2e7),
        ("dkd_dipole", "Dipole", "drift_kick_drift", dict(gap_exit=0.04, **dip), dict(fringe_at="both"), 6e7),
        ("dkd_dipole_entrance", "Dipole", "drift_kick_drift", dict(gap_exit=0.04, **dip), dict(fringe_at="entrance"), 6e7),
        ("dkd_tdc", "TransverseDeflectingCavity", "drift_kick_drift",
         dict(length=1.0, voltage=1e7, phase=0.2, frequency=1e9, tilt=0.05, misalignment=[2e-4, -1e-4]), {}, 1e8),
        ("so_drift", "Drift", "second_order", dict(length=0.8), {}, 1e8),
        ("so_quad", "Quadrupole", "second_order", dict(length=0.3, k1=4.2, tilt=0.3, misalignment=[1e-3, -2e-3]), {}, 1e8),
        ("so_quad_defocus", "Quadrupole", "second_order", dict(length=0.2, k1=-7.0), {}, 2e7),
        ("so_dipole", "Dipole", "second_order", dict(k1=0.7, **dip), {}, 6e7),
        ("so_sextupole", "Sextupole", "second_order", dict(length=0.4, k2=12.0, tilt=0.2, misalignment=[1e-3, 5e-4]), {}, 1e8),
    ]
    for name, cls, method, params, opts, E in cases:
        tensors = {k: torch.nn.Parameter(t64(v)) for k, v in params.items()}
        energy = t64(E).requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        el = getattr(cheetah, cls)(**tensors, **opts, tracking_method=method, **F64)
        beam = cheetah.ParticleBeam(xin, energy, species=cheetah.Species("electron", **F64), **F64)
        out = el.track(beam)
        loss = (out.particles * W).sum() + 1e-9 * out.energy.sum()
        loss.backward()
        arrays[f"{name}__class"] = np.asarray(cls)
        arrays[f"{name}__method"] = np.asarray(method)
        arrays[f"{name}__opts"] = np.asarray(repr(opts))
        arrays[f"{name}__energy"] = np.asarray(E)
        arrays[f"{name}__pnames"] = np.asarray(list(params))
        for k, tns in tensors.items():
            arrays[f"{name}__p__{k}"] = npy(tns.detach())
            arrays[f"{name}__g__{k}"] = npy(tns.grad)
        arrays[f"{name}__g_energy"] = npy(energy.grad)
        arrays[f"{name}__dx"] = npy(xin.grad)
        arrays[f"{name}__out"] = npy(out.particles.detach())
        arrays[f"{name}__loss"] = npy(loss.detach())
        print(name, float(loss), {k: npy(v.grad).tolist() for k, v in tensors.items()}, float(energy.grad))
    arrays["names"] = np.asarray([c[0] for c in cases])
    save("nonlinear_grad.npz", **arrays)


def gen_parameter_cavity_grad():
    """Gradients through Cavity.track for a ParameterBeam (cavity.py:108-110, 127-133, 202-218): d loss / d (voltage,
    phase, frequency, length, incoming energy, mu, cov), loss = sum(Wm * mu') + 1e6 sum(Wc * cov') + 1e-15 * energy', fp64."""
    arrays = {}
    g = torch.Generator().manual_seed(17)
    Wm, Wc = torch.randn(7, generator=g, **F64), torch.randn(7, 7, generator=g, **F64)
    arrays["Wm"], arrays["Wc"] = npy(Wm), npy(Wc)
    base = cheetah.ParameterBeam.from_parameters(sigma_x=t64(2e-4), sigma_px=t64(4e-6), sigma_y=t64(3e-4),
                                                 sigma_py=t64(5e-6), sigma_tau=t64(8e-6), sigma_p=t64(2e-3),
                                                 mu_x=t64(1e-5), mu_tau=t64(2e-6), mu_p=t64(1e-4), cov_taup=t64(3e-9),
                                                 energy=t64(6e6), **F64)
    arrays["mu"], arrays["cov"] = npy(base.mu), npy(base.cov)
    cases = [("sw_acc", "standing_wave", 1.0377, 18.15975e6, 30.0, 1.3e9, 6e6),
             ("tw_acc", "traveling_wave", 4.139, 2.0e7, -12.0, 2.998e9, 1e8),
             ("sw_dec", "standing_wave", 1.0377, 5.0e6, 170.0, 1.3e9, 5e7)]
    for name, ctype, L, V, ph, f, E in cases:
        length = torch.nn.Parameter(t64(L))
        voltage, phase, freq = torch.nn.Parameter(t64(V)), torch.nn.Parameter(t64(ph)), torch.nn.Parameter(t64(f))
        energy = t64(E).requires_grad_(True)
        mu = base.mu.clone().requires_grad_(True)
        cov = base.cov.clone().requires_grad_(True)
        cav = cheetah.Cavity(length=length, voltage=voltage, phase=phase, frequency=freq, cavity_type=ctype, **F64)
        beam = cheetah.ParameterBeam(mu, cov, energy, species=cheetah.Species("electron", **F64), **F64)
        out = cav.track(beam)
        loss = (out.mu * Wm).sum() + 1e6 * (out.cov * Wc).sum() + 1e-15 * out.energy
        loss.backward()
        arrays[f"{name}_params"] = np.asarray([L, V, ph, f, E])
        arrays[f"{name}_type"] = np.asarray(ctype)
        arrays[f"{name}_loss"] = npy(loss.detach())
        arrays[f"{name}_mu_out"], arrays[f"{name}_cov_out"] = npy(out.mu.detach()), npy(out.cov.detach())
        arrays[f"{name}_grads"] = np.asarray([float(length.grad), float(voltage.grad), float(phase.grad), float(freq.grad),
                                              float(energy.grad)])
        arrays[f"{name}_dmu"], arrays[f"{name}_dcov"] = npy(mu.grad), npy(cov.grad)
        print(name, float(loss.detach()), arrays[f"{name}_grads"])
    arrays["names"] = np.asarray([c[0] for c in cases])
    save("parameter_cavity_grad.npz", **arrays)


def gen_sc_grad():
    """Gradients through SpaceChargeKick.track (space_charge_kick.py:497-590, differentiated by torch autograd; the
    reference's own tests/test_space_charge_kick.py:202-327 rely on it): d loss / d (effect_length, incoming energy,
    incoming particles, particle charges), loss = sum(W * outgoing particles), fp64, 16^3 grid."""
    arrays = {}
    g = torch.Generator().manual_seed(23)
    N = 3000
    x = torch.randn(N, 7, generator=g, **F64) * t64([3e-4, 2e-5, 2e-4, 3e-5, 1e-5, 1e-3, 0.0])
    x[:, 6] = 1.0
    W = torch.randn(N, 7, generator=g, **F64)
    q0 = torch.full((N,), 2e-9 / N, **F64) * (1.0 + 0.2 * torch.rand(N, generator=g, **F64))
    arrays["x"], arrays["W"], arrays["q"] = npy(x), npy(W), npy(q0)
    cases = [("e50MeV", 0.3, 5e7, (16, 16, 16), 3.0), ("e8MeV_anis", 0.1, 8e6, (16, 32, 16), 2.5)]
    for name, L, E, grid, ext in cases:
        length = torch.nn.Parameter(t64(L))
        energy = t64(E).requires_grad_(True)
        xin = x.clone().requires_grad_(True)
        q = q0.clone().requires_grad_(True)
        sc = cheetah.SpaceChargeKick(effect_length=length, grid_shape=grid, grid_extent_x=t64(ext), grid_extent_y=t64(ext),
                                     grid_extent_tau=t64(ext), **F64)
        beam = cheetah.ParticleBeam(xin, energy, particle_charges=q, species=cheetah.Species("electron", **F64), **F64)
        out = sc.track(beam)
        loss = (out.particles * W).sum()
        loss.backward()
        arrays[f"{name}_meta"] = np.asarray([L, E, ext, *grid])
        arrays[f"{name}_out"] = npy(out.particles.detach())
        arrays[f"{name}_loss"] = npy(loss.detach())
        arrays[f"{name}_grads"] = np.asarray([float(length.grad), float(energy.grad)])
        arrays[f"{name}_dx"] = npy(xin.grad)
        arrays[f"{name}_dq"] = npy(q.grad)
        print(name, float(loss.detach()), arrays[f"{name}_grads"], float(xin.grad.abs().max()), float(q.grad.abs().max()))
    arrays["names"] = np.asarray([c[0] for c in cases])
    save("sc_grad.npz", **arrays)


def gen_kde():
    """Screen(method="kde") images (screen.py:312-326, utils/kde.py): scalar and vectorised beams, misalignment,
    survival weights; fp64 and fp32; plus d(image)/d(particles) through a weighted sum for the fp64 scalar case."""
    arrays = {}
    g = torch.Generator().manual_seed(31)
    N = 1500
    x = torch.randn(2, N, 7, generator=g, **F64) * t64([4e-4, 1e-5, 3e-4, 1e-5, 1e-5, 1e-3, 0.0])
    x[..., 6] = 1.0
    q = (torch.rand(N, generator=g, **F64) + 0.5) * 1e-15 * torch.where(torch.rand(N, generator=g, **F64) < 0.3, -1.0, 1.0)
    surv = torch.rand(N, generator=g, **F64)
    W = torch.randn(48, 64, generator=g, **F64)
    arrays.update(x=npy(x), q=npy(q), surv=npy(surv), W=npy(W))
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        kw = {"dtype": dtype}
        scr = cheetah.Screen(resolution=(64, 48), pixel_size=torch.tensor([4e-5, 5e-5], **kw), method="kde",
                             kde_bandwidth=torch.tensor(6e-5, **kw), misalignment=torch.tensor([1e-4, -5e-5], **kw),
                             is_active=True, **kw)
        for name, parts in (("scalar", x[0]), ("vector", x)):
            xin = parts.to(dtype).clone().requires_grad_(dtype == torch.float64 and name == "scalar")
            beam = cheetah.ParticleBeam(xin, torch.tensor(1e8, **kw), particle_charges=q.to(dtype),
                                        survival_probabilities=surv.to(dtype), **kw)
            scr.track(beam)
            img = scr.reading
            arrays[f"{name}_{tag}"] = npy(img.detach())
            if xin.requires_grad:
                (img * W).sum().backward()
                arrays["scalar_f64_dx"] = npy(xin.grad)
            print(name, tag, tuple(img.shape), float(img.sum()))
    save("kde.npz", **arrays)


def gen_special():
    """Values and gradients of the reference's special functions (utils/autograd.py) at regular points, at the removed
    singularities (0, a == b, b == 0) and for negative arguments, plus utils.kde / utils.bmadx round-trip data."""
    from cheetah.utils import autograd as ag
    from cheetah.utils import bmadx, kde_histogram_1d, kde_histogram_2d

    arrays = {}
    x = torch.tensor([-30.0, -2.5, -0.5, -1e-3, 0.0, 1e-3, 0.3, 1.0, 9.8696, 42.0], dtype=torch.float64)
    arrays["x"] = npy(x)
    for name in ("log1pdiv", "si1mdiv", "sicos1mdiv", "sipsicos3mdiv", "sicoskuddelmuddel15mdiv"):
        xin = (x.clamp_min(-0.9) if name == "log1pdiv" else x).clone().requires_grad_(True)
        y = getattr(ag, name)(xin)
        (g,) = torch.autograd.grad(y.sum(), xin)
        arrays[name + "_x"], arrays[name], arrays[name + "_grad"] = npy(xin.detach()), npy(y.detach()), npy(g)
    a = torch.tensor([-3.0, -0.5, 0.0, 0.0, 0.7, 0.7, 2.0, 5.0, 12.0, -1.5, 4.0], dtype=torch.float64)
    b = torch.tensor([-3.0, 0.25, 0.0, 1.5, 0.7, -0.7, 3.0, 5.0, 1.0, -1.5, 0.0], dtype=torch.float64)
    arrays["a"], arrays["b"] = npy(a), npy(b)
    for name in ("cossqrtmcosdivdiff", "simsidivdiff", "si2msi2divdiff"):
        ain, bin_ = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = getattr(ag, name)(ain, bin_)
        ga, gb = torch.autograd.grad(y.sum(), (ain, bin_))
        arrays[name], arrays[name + "_ga"], arrays[name + "_gb"] = npy(y.detach()), npy(ga), npy(gb)
    a2 = torch.tensor([0.5, 1.0, 2.0, 3.0, 1.5], dtype=torch.float64, requires_grad=True)
    b2 = torch.tensor([0.0, 0.3, -1.0, 0.0, 10.0], dtype=torch.float64, requires_grad=True)
    y = ag.sqrta2minusbdiva(a2, b2)
    ga, gb = torch.autograd.grad(y.sum(), (a2, b2))
    arrays.update(sq_a=npy(a2.detach()), sq_b=npy(b2.detach()), sqrta2minusbdiva=npy(y.detach()), sqrta2minusbdiva_ga=npy(ga),
                  sqrta2minusbdiva_gb=npy(gb))
    # utils.kde with the reference's call signatures
    torch.manual_seed(5)
    s1, s2 = torch.randn(2, 300, dtype=torch.float64), 0.5 * torch.randn(2, 300, dtype=torch.float64) + 0.2
    wts = torch.rand(2, 300, dtype=torch.float64)
    bins1, bins2 = torch.linspace(-3, 3, 40, dtype=torch.float64), torch.linspace(-2, 2, 25, dtype=torch.float64)
    bw = torch.tensor(0.15, dtype=torch.float64)
    arrays.update(kde_s1=npy(s1), kde_s2=npy(s2), kde_w=npy(wts), kde_bins1=npy(bins1), kde_bins2=npy(bins2),
                  kde_1d=npy(kde_histogram_1d(s1, bins1, bw)), kde_1d_w=npy(kde_histogram_1d(s1, bins1, bw, weights=wts)),
                  kde_2d=npy(kde_histogram_2d(s1, s2, bins1, bins2, bw)),
                  kde_2d_w=npy(kde_histogram_2d(s1, s2, bins1, bins2, bw, weights=wts)))
    # utils.bmadx coordinate conversions
    coords = torch.randn(2, 50, 7, dtype=torch.float64) * torch.tensor([1e-3, 1e-4, 1e-3, 1e-4, 1e-3, 1e-3, 0.0], dtype=torch.float64)
    coords[..., 6] = 1.0
    ref_energy, mc2 = torch.tensor([5e6, 1e8], dtype=torch.float64), torch.tensor(510998.95, dtype=torch.float64)
    bm, p0c = bmadx.cheetah_to_bmad_coords(coords, ref_energy, mc2)
    back, e_back = bmadx.bmad_to_cheetah_coords(bm, p0c, mc2)
    arrays.update(bx_coords=npy(coords), bx_ref_energy=npy(ref_energy), bx_mc2=npy(mc2), bx_bmad=npy(bm), bx_p0c=npy(p0c),
                  bx_back=npy(back), bx_e_back=npy(e_back))
    save("special.npz", **arrays)
    print({k: v.shape for k, v in arrays.items() if k.endswith("_grad") or k in ("kde_2d", "bx_bmad")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["special", "dkd", "second_order", "aperture", "ares", "beam_utils", "cavity_grad", "nonlinear_grad", "parameter_cavity_grad", "sc_grad", "kde"]
    for w in which:
        globals()["gen_" + w]()

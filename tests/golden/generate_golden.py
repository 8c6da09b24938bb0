#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (it imports /root/reference read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden.py

Everything written is DATA: inputs and the reference's outputs as plain numpy arrays
(.npz).  No reference source, bytecode or pickled reference objects are stored.  The
reference's own pickled goldens (tests/resources/consistency_expected_outgoing/*.pkl,
tests/test_elements.py:356-431) are re-exported as arrays.

Environment used for the committed files: torch 2.10.0+rocm7.0 (CPU), scipy 1.15.3
(CODATA 2022: m_e = 510998.95069 eV), numpy 2.2.6, reference v0.8.5-dev.
"""

import json
import os
import pickle
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cheetah  # noqa: E402  (the reference)
from cheetah.utils.cloud_in_cell import cloud_in_cell_charge_deposition  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def species_meta(sp):
    return float(sp.mass_eV), float(sp.num_elementary_charges)


# ----------------------------------------------------------------------------------------------
def gen_maps():
    """First-order maps of every hot-path element (rows a3-a9), fp64, scalar and vectorised."""
    cases = []
    f64 = {"dtype": torch.float64}
    t = lambda v: torch.tensor(v, **f64)  # noqa: E731
    energies = [6e6, 1e8, 1.0732e8, 14e9]
    elec = cheetah.Species("electron", **f64)
    prot = cheetah.Species("proton", **f64)

    def add(kind, elem, params, energy, species):
        R = elem.first_order_transfer_map(t(energy), species)
        m, q = species_meta(species)
        cases.append({"kind": kind, "params": np.asarray(params, dtype=np.float64),
                      "energy": np.asarray(energy, dtype=np.float64), "mass_eV": m, "n_charges": q,
                      "R": npy(R)})

    for E in energies:
        for sp in (elec, prot):
            if sp is prot and E < 2e9:
                continue
            for L in (1.0, 0.2, -1.0):
                add("drift", cheetah.Drift(length=t(L), **f64), [L], E, sp)
            for (L, k1, tilt, mx, my) in [(0.2, 4.2, 0.0, 0.0, 0.0), (0.2, -4.2, 0.0, 0.0, 0.0),
                                          (1.0, 0.0, 0.0, 0.0, 0.0), (1.0, 1.0, 0.42, 0.01, -0.02),
                                          (1.0, -2.0, 0.42, 0.01, -0.02), (0.122, 8.2, 0.0, 0.0, 0.0),
                                          (0.5, 1e-9, 0.1, 0.0, 0.0), (0.3, -30.0, 0.0, 1e-3, 2e-3)]:
                add("quadrupole",
                    cheetah.Quadrupole(length=t(L), k1=t(k1), tilt=t(tilt), misalignment=t([mx, my]), **f64),
                    [L, k1, tilt, mx, my], E, sp)
            for (L, ang, k1, e1, e2, tilt, fint, fintx, gap) in [
                (1.0, 0.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
                (1.0, 1.0, 0.0, 0.0, 0.0, 0.42, 0.0, 0.0, 0.0),
                (1.0, -2.0, 0.0, 0.0, 0.0, 0.42, 0.0, 0.0, 0.0),
                (0.5, 0.2, 1.5, 0.05, -0.03, 0.1, 0.5, 0.4, 0.02),
                (0.5, 0.2, -3.0, 0.1, 0.1, 0.0, 0.3, 0.3, 0.03),
                (1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
                (0.7, 1e-5, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0),
            ]:
                add("dipole",
                    cheetah.Dipole(length=t(L), angle=t(ang), k1=t(k1), dipole_e1=t(e1), dipole_e2=t(e2),
                                   tilt=t(tilt), fringe_integral=t(fint), fringe_integral_exit=t(fintx),
                                   gap=t(gap), **f64),
                    [L, ang, k1, e1, e2, tilt, fint, fintx, gap], E, sp)
            add("hcor", cheetah.HorizontalCorrector(length=t(0.02), angle=t(9e-5), **f64), [0.02, 9e-5], E, sp)
            add("vcor", cheetah.VerticalCorrector(length=t(0.02), angle=t(-1e-4), **f64), [0.02, -1e-4], E, sp)
            add("ccor", cheetah.CombinedCorrector(length=t(1.0), horizontal_angle=t(1.0), vertical_angle=t(-2.0), **f64),
                [1.0, 1.0, -2.0], E, sp)
            add("identity", cheetah.Marker(**f64), [], E, sp)
            for ctype, kind in (("standing_wave", "cavity_sw"), ("traveling_wave", "cavity_tw")):
                for (L, V, ph, fr) in [(1.0377, 18.15975e6, 30.0, 1.3e9), (1.0377, 18.15975e6, 0.0, 1.3e9),
                                       (1.0377, 18.15975e6, -60.0, 1.3e9), (1.0, 0.0, 0.0, 0.0),
                                       (1.0377, 18.15975e6, 90.0, 1.3e9), (0.5, -5e6, 10.0, 2.998e9)]:
                    if E + 2e7 < 0 or (E < 3e7 and V < 0):
                        continue
                    add(kind, cheetah.Cavity(length=t(L), voltage=t(V), phase=t(ph), frequency=t(fr),
                                             cavity_type=ctype, **f64), [L, V, ph, fr], E, sp)
    # vectorised (3,2) broadcast: k1 (3,1) x tilt (2,) (tests/test_vectorized.py:145-183 pattern)
    k1 = t([[4.2], [-4.2], [0.0]])
    tilt = t([0.0, 0.3])
    q = cheetah.Quadrupole(length=t(0.2), k1=k1, tilt=tilt, **f64)
    Rv = q.first_order_transfer_map(t(1e8), elec)
    pv = np.zeros((3, 2, 5))
    pv[..., 0] = 0.2
    pv[..., 1] = npy(k1)
    pv[..., 2] = npy(tilt)
    cases.append({"kind": "quadrupole", "params": pv.reshape(6, 5), "energy": np.asarray(1e8),
                  "mass_eV": species_meta(elec)[0], "n_charges": -1.0, "R": npy(Rv).reshape(6, 7, 7)})
    arrays = {"n_cases": np.asarray(len(cases))}
    for i, c in enumerate(cases):
        arrays[f"kind_{i}"] = np.asarray(c["kind"])
        arrays[f"params_{i}"] = c["params"]
        arrays[f"energy_{i}"] = c["energy"]
        arrays[f"species_{i}"] = np.asarray([c["mass_eV"], c["n_charges"]])
        arrays[f"R_{i}"] = c["R"]
    save("maps.npz", **arrays)


# ----------------------------------------------------------------------------------------------
def load_incoming():
    with open(os.path.join(REF, "tests/resources/ACHIP_EA1_2021.1351.001_subsampled_3000.pkl"), "rb") as f:
        return pickle.load(f)


def gen_consistency():
    """Re-export of the reference's own pickled goldens (tests/test_elements.py:356-431)."""
    incoming32 = load_incoming()
    incoming = incoming32.to(torch.float64)
    arrays = {
        "incoming_particles_f32": npy(incoming32.particles),
        "incoming_energy": npy(incoming.energy),
        "incoming_charges_f32": npy(incoming32.particle_charges),
        "incoming_survival_f32": npy(incoming32.survival_probabilities),
        "incoming_s": npy(incoming.s),
        "species": np.asarray(species_meta(incoming.species)),
    }
    wanted = ["Drift_ParticleBeam_linear", "Quadrupole_ParticleBeam_linear", "Dipole_ParticleBeam_linear",
              "RBend_ParticleBeam_linear", "HorizontalCorrector_ParticleBeam_default",
              "VerticalCorrector_ParticleBeam_default", "CombinedCorrector_ParticleBeam_default",
              "Cavity_ParticleBeam_default", "CustomTransferMap_ParticleBeam_identity",
              "Marker_ParticleBeam_default", "Screen_ParticleBeam_default", "Segment_ParticleBeam_default",
              "SpaceChargeKick_ParticleBeam_default", "BPM_ParticleBeam_inactive",
              "Aperture_ParticleBeam_inactive"]
    keep = 512  # particles kept per expected-outgoing array (size)
    for name in wanted:
        with open(os.path.join(REF, "tests/resources/consistency_expected_outgoing", name + ".pkl"), "rb") as f:
            exp = pickle.load(f)
        arrays[f"{name}__particles"] = npy(exp.particles)[..., :keep, :]
        arrays[f"{name}__energy"] = npy(exp.energy)
        arrays[f"{name}__s"] = npy(exp.s)
    arrays["keep"] = np.asarray(keep)
    save("consistency.npz", **arrays)


# ----------------------------------------------------------------------------------------------
def readme_segment(dtype):
    f = {"dtype": dtype}
    t = lambda v: torch.tensor(v, **f)  # noqa: E731
    seg = cheetah.Segment(elements=[
        cheetah.Drift(length=t(0.175), **f),
        cheetah.Quadrupole(length=t(0.122), name="AREAMQZM1", **f),
        cheetah.Drift(length=t(0.428), **f),
        cheetah.Quadrupole(length=t(0.122), name="AREAMQZM2", **f),
        cheetah.Drift(length=t(0.204), **f),
        cheetah.VerticalCorrector(length=t(0.02), name="AREAMCVM1", **f),
        cheetah.Drift(length=t(0.204), **f),
        cheetah.Quadrupole(length=t(0.122), name="AREAMQZM3", **f),
        cheetah.Drift(length=t(0.179), **f),
        cheetah.HorizontalCorrector(length=t(0.02), name="AREAMCHM1", **f),
        cheetah.Drift(length=t(0.45), **f),
        cheetah.Screen(name="AREABSCR1", **f),
    ])
    seg.AREAMQZM1.k1 = t(8.2)
    seg.AREAMQZM2.k1 = t(-14.3)
    seg.AREAMCVM1.angle = t(9e-5)
    seg.AREAMQZM3.k1 = t(3.142)
    seg.AREAMCHM1.angle = t(-1e-4)
    return seg


MOMENT_NAMES = ["mu_x", "mu_px", "mu_y", "mu_py", "mu_tau", "mu_p", "sigma_x", "sigma_px", "sigma_y",
                "sigma_py", "sigma_tau", "sigma_p", "cov_xpx", "cov_ypy", "cov_taup", "total_charge"]


def gen_segment_readme():
    """C1: README 12-element ARES segment, from_twiss beam, fp64 (README.md:43-88)."""
    torch.manual_seed(1234)
    N = 4096
    beam = cheetah.ParticleBeam.from_twiss(beta_x=torch.tensor(3.14, dtype=torch.float64),
                                           beta_y=torch.tensor(42.0, dtype=torch.float64),
                                           num_particles=N, dtype=torch.float64).to(torch.float64)
    seg = readme_segment(torch.float64)
    seg.AREABSCR1.is_active = True
    seg.AREABSCR1.pixel_size = torch.tensor([1e-5, 1e-5], dtype=torch.float64)
    seg.AREABSCR1.resolution = (512, 256)
    out = seg.track(beam)
    R = seg.first_order_transfer_map(beam.energy, beam.species) if seg.is_skippable else None
    img = seg.AREABSCR1.reading
    nz = torch.nonzero(img)
    arrays = {"in_particles": npy(beam.particles), "energy": npy(beam.energy), "charges": npy(beam.particle_charges),
              "survival": npy(beam.survival_probabilities), "out_particles": npy(out.particles),
              "out_s": npy(out.s), "img_shape": np.asarray(img.shape), "img_idx": npy(nz).astype(np.int32),
              "img_val": npy(img[nz[:, 0], nz[:, 1]]), "resolution": np.asarray([512, 256]),
              "pixel_size": np.asarray([1e-5, 1e-5])}
    seg2 = readme_segment(torch.float64)  # screen inactive -> all skippable
    arrays["R_merged"] = npy(seg2.first_order_transfer_map(beam.energy, beam.species))
    for n in MOMENT_NAMES:
        arrays["mom_" + n] = npy(getattr(out, n))
    del R
    save("segment_readme.npz", **arrays)


def fodo_segment(dtype, cells=25):
    f = {"dtype": dtype}
    t = lambda v: torch.tensor(v, **f)  # noqa: E731
    els = []
    for _ in range(cells):
        els += [cheetah.Quadrupole(length=t(0.2), k1=t(4.2), **f), cheetah.Drift(length=t(0.8), **f),
                cheetah.Quadrupole(length=t(0.2), k1=t(-4.2), **f), cheetah.Drift(length=t(0.8), **f)]
    return cheetah.Segment(elements=els)


def gen_fodo100():
    """C2: 100-element FODO, fp32 and fp64, merged map + element-by-element tracking."""
    torch.manual_seed(1234)
    N = 4096
    arrays = {}
    beam32 = cheetah.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float32)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        beam = beam32.to(dt)
        seg = fodo_segment(dt)
        out = seg.track(beam)
        b = beam
        for e in seg.elements:
            b = e.track(b)
        arrays[f"in_{tag}"] = npy(beam.particles)
        arrays[f"R_{tag}"] = npy(seg.first_order_transfer_map(beam.energy, beam.species))
        arrays[f"out_merged_{tag}"] = npy(out.particles)
        arrays[f"out_elementwise_{tag}"] = npy(b.particles)
        arrays[f"energy_{tag}"] = npy(beam.energy)
        arrays[f"species_{tag}"] = np.asarray(species_meta(beam.species))  # NB: fp32-rounded mass (beam32.to)
    save("fodo100.npz", **arrays)


def ares_subcell(dtype, k1_scan):
    """ARES EA subcell AREASOLA1 -> AREABSCR1 (docs/examples/ARESlatticeStage3v1_9.json), 13 elements."""
    f = {"dtype": dtype}
    t = lambda v: torch.tensor(v, **f)  # noqa: E731
    return cheetah.Segment(elements=[
        cheetah.Marker(name="AREASOLA1", **f),
        cheetah.Drift(length=t(0.17504), **f),
        cheetah.Quadrupole(length=t(0.122), k1=k1_scan, name="AREAMQZM1", **f),
        cheetah.Drift(length=t(0.428), **f),
        cheetah.Quadrupole(length=t(0.122), k1=t(-14.3), name="AREAMQZM2", **f),
        cheetah.Drift(length=t(0.204), **f),
        cheetah.VerticalCorrector(length=t(0.02), angle=t(9e-5), name="AREAMCVM1", **f),
        cheetah.Drift(length=t(0.204), **f),
        cheetah.Quadrupole(length=t(0.122), k1=t(3.142), name="AREAMQZM3", **f),
        cheetah.Drift(length=t(0.179), **f),
        cheetah.HorizontalCorrector(length=t(0.02), angle=t(-1e-4), name="AREAMCHM1", **f),
        cheetah.Drift(length=t(0.45), **f),
        cheetah.Screen(resolution=(2448, 2040), pixel_size=t([3.5488e-6, 2.5003e-6]), name="AREABSCR1", **f),
    ])


def gen_k1scan():
    """C3 (reduced): k1 scan on the ARES EA subcell with a shared beam."""
    torch.manual_seed(1234)
    N, B = 2048, 16
    arrays = {}
    beam32 = cheetah.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float32)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        k1 = torch.linspace(-30, 30, B, dtype=dt)
        seg = ares_subcell(dt, k1)
        beam = beam32.to(dt)
        out = seg.track(beam)
        arrays[f"k1_{tag}"] = npy(k1)
        arrays[f"in_{tag}"] = npy(beam.particles)
        arrays[f"R_{tag}"] = npy(seg.first_order_transfer_map(beam.energy, beam.species))
        arrays[f"out_{tag}"] = npy(out.particles)[:, :256]
        arrays[f"sigma_x_{tag}"] = npy(out.sigma_x)
        arrays[f"sigma_y_{tag}"] = npy(out.sigma_y)
        arrays[f"mu_x_{tag}"] = npy(out.mu_x)
        arrays[f"energy_{tag}"] = npy(beam.energy)
        arrays[f"species_{tag}"] = np.asarray(species_meta(beam.species))
    save("k1scan.npz", **arrays)


def gen_cavity():
    """a8: Cavity.track (cavity.py:100-251), SW/TW x phases x voltages, vectorised voltage."""
    torch.manual_seed(4321)
    N = 128
    arrays = {}
    i = 0
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        t = lambda v: torch.tensor(v, dtype=dt)  # noqa: E731
        for E in (6e6, 1e8):
            beam = cheetah.ParticleBeam.from_parameters(num_particles=N, energy=t(E), sigma_tau=t(1e-4),
                                                        sigma_p=t(1e-3), dtype=dt).to(dt)
            for ctype in ("standing_wave", "traveling_wave"):
                for (L, V, ph, fr) in [(1.0377, 18.15975e6, 30.0, 1.3e9), (1.0377, 18.15975e6, 0.0, 1.3e9),
                                       (1.0377, 18.15975e6, -60.0, 1.3e9), (1.0377, -1.0e6, 20.0, 1.3e9),
                                       (1.0377, [18.15975e6, -2.0e6, 5.0e6], 30.0, 1.3e9)]:
                    cav = cheetah.Cavity(length=t(L), voltage=t(V), phase=t(ph), frequency=t(fr),
                                         cavity_type=ctype, dtype=dt)
                    out = cav.track(beam)
                    arrays[f"c{i}_meta"] = np.asarray(json.dumps({"dtype": tag, "type": ctype, "E": E}))
                    arrays[f"c{i}_params"] = np.stack(np.broadcast_arrays(np.float64(L), np.asarray(V, dtype=np.float64),
                                                                          np.float64(ph), np.float64(fr)), axis=-1).reshape(-1, 4)
                    arrays[f"c{i}_in"] = npy(beam.particles)
                    arrays[f"c{i}_out"] = npy(out.particles)
                    arrays[f"c{i}_energy_out"] = npy(out.energy)
                    arrays[f"c{i}_R"] = npy(cav.first_order_transfer_map(beam.energy, beam.species))
                    i += 1
    arrays["n_cases"] = np.asarray(i)
    save("cavity.npz", **arrays)


def gen_moments():
    """a10: weighted moments with non-trivial survival probabilities, offset beam, vectorised."""
    torch.manual_seed(99)
    arrays = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        N = 5000
        beam = cheetah.ParticleBeam.from_parameters(num_particles=N, mu_x=torch.tensor(2e-3, dtype=dt),
                                                    mu_py=torch.tensor(-1e-4, dtype=dt),
                                                    cov_xpx=torch.tensor(2e-10, dtype=dt), dtype=dt).to(dt)
        parts = beam.particles.unsqueeze(0).repeat(3, 1, 1)
        parts[1] *= 1.5
        parts[2, :, 0] += 1e-2
        surv = torch.rand(3, N, dtype=dt)
        surv[0] = 1.0
        vb = cheetah.ParticleBeam(parts, beam.energy, particle_charges=beam.particle_charges,
                                  survival_probabilities=surv, dtype=dt)
        arrays[f"particles_{tag}"] = npy(parts)
        arrays[f"survival_{tag}"] = npy(surv)
        arrays[f"charges_{tag}"] = npy(beam.particle_charges)
        names = ["mu_x", "mu_px", "mu_y", "mu_py", "mu_tau", "mu_p", "sigma_x", "sigma_px", "sigma_y",
                 "sigma_py", "sigma_tau", "sigma_p", "total_charge"] + \
                [n for n in dir(vb) if n.startswith("cov_")]
        for n in names:
            arrays[f"{n}_{tag}"] = npy(getattr(vb, n))
        for n in ["emittance_x", "emittance_y", "beta_x", "beta_y", "alpha_x", "alpha_y",
                  "normalized_emittance_x", "relativistic_gamma", "relativistic_beta"]:
            arrays[f"{n}_{tag}"] = npy(getattr(vb, n))
    save("moments.npz", **arrays)


def gen_cic():
    """a12: cloud_in_cell_charge_deposition 1/2/3-D incl. out-of-extent particles (cloud_in_cell.py)."""
    torch.manual_seed(7)
    arrays = {}
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        N = 3000
        pos = torch.randn(2, N, 3, dtype=dt) * torch.tensor([1e-3, 2e-3, 5e-4], dtype=dt)
        q = torch.rand(2, N, dtype=dt) * 1e-15
        ext = torch.tensor([[-2e-3, 2.5e-3], [-3e-3, 3e-3], [-1e-3, 8e-4]], dtype=dt)
        # exact edge / just outside cases
        pos[0, 0] = ext[:, 0]
        pos[0, 1] = ext[:, 1]
        pos[0, 2] = torch.nextafter(ext[:, 1], torch.tensor(1.0, dtype=dt))
        pos[0, 3] = torch.nextafter(ext[:, 0], torch.tensor(-1.0, dtype=dt))
        arrays[f"pos_{tag}"] = npy(pos)
        arrays[f"q_{tag}"] = npy(q)
        arrays[f"ext_{tag}"] = npy(ext)
        for nd, bins in ((1, (17,)), (2, (16, 12)), (3, (8, 6, 10))):
            g = cloud_in_cell_charge_deposition(pos[..., :nd], bins=list(bins), extent=ext[:nd], charges=q)
            arrays[f"grid{nd}d_{tag}"] = npy(g)
        # vectorised extent (per batch row)
        extv = torch.stack([ext[:2], ext[:2] * 0.5])
        g = cloud_in_cell_charge_deposition(pos[..., :2], bins=[9, 7], extent=extv, charges=q)
        arrays[f"ext_v_{tag}"] = npy(extv)
        arrays[f"grid2d_v_{tag}"] = npy(g)
    save("cic.npz", **arrays)


def gen_screen():
    """a11: Screen histogram / CIC readings and bit-exact pixel indices (screen.py:139-344)."""
    torch.manual_seed(11)
    arrays = {}
    i = 0
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for (res, px, binning, mis) in [((64, 48), (1e-4, 1.5e-4), 1, (0.0, 0.0)),
                                        ((64, 48), (1e-4, 1.5e-4), 2, (3e-4, -2e-4)),
                                        ((2448, 2040), (3.3198e-6, 2.4469e-6), 1, (0.0, 0.0)),
                                        ((1024, 1024), (1e-3, 1e-3), 1, (0.0, 0.0))]:
            N = 2000
            half_w, half_h = res[0] * px[0] / 2, res[1] * px[1] / 2
            parts = torch.zeros(N, 7, dtype=dt)
            parts[:, 0] = torch.randn(N, dtype=dt) * half_w * 0.6
            parts[:, 2] = torch.randn(N, dtype=dt) * half_h * 0.6
            parts[:, 6] = 1.0
            q = -(torch.rand(N, dtype=dt) + 0.5) * 1e-15
            surv = torch.rand(N, dtype=dt)
            for method in ("histogram", "cloud-in-cell"):
                scr = cheetah.Screen(resolution=res, pixel_size=torch.tensor(px, dtype=dt), binning=binning,
                                     misalignment=torch.tensor(mis, dtype=dt), method=method, is_active=True,
                                     dtype=dt)
                if method == "histogram":
                    ex, ey = scr.pixel_bin_edges
                    # exact-edge and +-1 ulp particles
                    k = 0
                    for e_arr, col in ((ex, 0), (ey, 2)):
                        for j in (0, 1, len(e_arr) // 2, len(e_arr) - 2, len(e_arr) - 1):
                            for shift in (0, 1, -1):
                                v = e_arr[j] + torch.tensor(mis[0 if col == 0 else 1], dtype=dt)
                                if shift == 1:
                                    v = torch.nextafter(v, torch.tensor(float("inf"), dtype=dt))
                                if shift == -1:
                                    v = torch.nextafter(v, torch.tensor(float("-inf"), dtype=dt))
                                parts[k, col] = v
                                k += 1
                beam = cheetah.ParticleBeam(parts.clone(), torch.tensor(1e8, dtype=dt), particle_charges=q,
                                            survival_probabilities=surv, dtype=dt)
                scr.track(beam)
                img = scr.reading
                nzi = torch.nonzero(img)
                arrays[f"s{i}_meta"] = np.asarray(json.dumps({"dtype": tag, "resolution": res, "pixel_size": px,
                                                              "binning": binning, "misalignment": mis, "method": method}))
                arrays[f"s{i}_particles"] = npy(parts)
                arrays[f"s{i}_q"] = npy(q)
                arrays[f"s{i}_surv"] = npy(surv)
                arrays[f"s{i}_img_idx"] = npy(nzi).astype(np.int32)
                arrays[f"s{i}_img_val"] = npy(img[nzi[:, 0], nzi[:, 1]])
                arrays[f"s{i}_img_shape"] = np.asarray(img.shape)
                arrays[f"s{i}_extent"] = npy(scr.extent)
                if method == "histogram":
                    ex, ey = scr.pixel_bin_edges
                    arrays[f"s{i}_edges_x"] = npy(ex)
                    arrays[f"s{i}_edges_y"] = npy(ey)
                    rb = scr.get_read_beam()
                    # per-particle bin index via one-particle histogramdd would be slow: recover it by
                    # searchsorted with torch.histogramdd's convention and VERIFY against the image
                    x, y = rb.x.contiguous(), rb.y.contiguous()
                    jx = torch.searchsorted(ex, x, right=True) - 1
                    jy = torch.searchsorted(ey, y, right=True) - 1
                    jx[x == ex[-1]] = len(ex) - 2
                    jy[y == ey[-1]] = len(ey) - 2
                    bad = (x < ex[0]) | (x > ex[-1]) | (y < ey[0]) | (y > ey[-1])
                    jx[bad] = -1
                    jy[bad] = -1
                    chk = torch.zeros_like(img)
                    w = q.abs() * surv
                    ok = ~bad
                    chk.index_put_((jy[ok], jx[ok]), w[ok], accumulate=True)
                    assert torch.allclose(chk, img, rtol=1e-5, atol=0), "histogramdd index convention check failed"
                    assert ((chk != 0) == (img != 0)).all()
                    arrays[f"s{i}_ij"] = np.stack([npy(jx), npy(jy)], axis=-1).astype(np.int32)
                i += 1
    arrays["n_cases"] = np.asarray(i)
    save("screen.npz", **arrays)


def gen_space_charge():
    """a13: SpaceChargeKick.track on a uniform ellipsoid, fp64 (+ fp32 outputs), 16^3 and 32^3 grids."""
    torch.manual_seed(2024)
    arrays = {}
    N = 4000
    for gi, grid in enumerate(((16, 16, 16), (32, 24, 20))):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            t = lambda v: torch.tensor(v, dtype=dt)  # noqa: E731
            torch.manual_seed(2024)
            beam = cheetah.ParticleBeam.uniform_3d_ellipsoid(
                num_particles=N, total_charge=t(1e-9), energy=t(2.5e8), radius_x=t(1e-3), radius_y=t(1e-3),
                radius_tau=t(1e-4), sigma_px=t(1e-5), sigma_py=t(1e-5), sigma_p=t(1e-5), dtype=dt).to(dt)
            surv = torch.ones(N, dtype=dt)
            surv[::7] = 0.5
            beam.survival_probabilities = surv
            sc = cheetah.SpaceChargeKick(effect_length=t(0.2), grid_shape=grid, dtype=dt)
            out = sc.track(beam)
            # intermediates (same calls as track, space_charge_kick.py:531-556)
            flat = cheetah.ParticleBeam(beam.particles.unsqueeze(0), beam.energy.unsqueeze(0),
                                        particle_charges=beam.particle_charges.unsqueeze(0),
                                        survival_probabilities=beam.survival_probabilities.unsqueeze(0), dtype=dt)
            half = torch.stack([sc.grid_extent_x * flat.sigma_x, sc.grid_extent_y * flat.sigma_y,
                                sc.grid_extent_tau * flat.sigma_tau], dim=-1)
            cell = 2 * half / torch.tensor(grid, dtype=dt)
            xp = flat.to_xyz_pxpypz()
            rho = sc._array_rho(flat, xp, cell, half)
            phi = sc._solve_poisson_equation(flat, xp, cell, half)
            forces = sc._compute_forces(flat, xp, cell, half)
            k = f"g{gi}_{tag}"
            arrays[f"{k}_in"] = npy(beam.particles)
            arrays[f"{k}_charges"] = npy(beam.particle_charges)
            arrays[f"{k}_survival"] = npy(beam.survival_probabilities)
            arrays[f"{k}_out"] = npy(out.particles)
            arrays[f"{k}_half"] = npy(half)
            arrays[f"{k}_cell"] = npy(cell)
            arrays[f"{k}_xp"] = npy(xp)[0, :256]
            arrays[f"{k}_rho"] = npy(rho)[0, : grid[0], : grid[1], : grid[2]]
            arrays[f"{k}_phi"] = npy(phi)[0]
            arrays[f"{k}_forces"] = npy(forces)[0, :512]
            arrays[f"{k}_grid"] = np.asarray(grid)
    arrays["energy"] = np.asarray(2.5e8)
    arrays["effect_length"] = np.asarray(0.2)
    save("space_charge.npz", **arrays)


def gen_grad():
    """C5: d sigma_x(screen) / d k1 through [Drift, Quad(k1), Drift, Screen], backward."""
    torch.manual_seed(1234)
    arrays = {}
    N = 4096
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        t = lambda v: torch.tensor(v, dtype=dt)  # noqa: E731
        torch.manual_seed(1234)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=N, dtype=torch.float32).to(dt)
        k1 = torch.nn.Parameter(t(3.142))
        L = torch.nn.Parameter(t(0.2))
        seg = cheetah.Segment(elements=[cheetah.Drift(length=t(1.0), dtype=dt),
                                        cheetah.Quadrupole(length=L, k1=k1, dtype=dt),
                                        cheetah.Drift(length=t(1.0), dtype=dt),
                                        cheetah.Screen(is_active=True, name="scr", dtype=dt)])
        parts = beam.particles.clone().requires_grad_(True)
        b = cheetah.ParticleBeam(parts, beam.energy, particle_charges=beam.particle_charges, dtype=dt)
        seg.track(b)
        rb = seg.scr.get_read_beam()
        loss = rb.sigma_x + 0.5 * rb.mu_y + 3.0 * rb.cov_xpx
        loss.backward()
        arrays[f"in_{tag}"] = npy(beam.particles)
        arrays[f"loss_{tag}"] = npy(loss)
        arrays[f"sigma_x_{tag}"] = npy(rb.sigma_x)
        arrays[f"dk1_{tag}"] = npy(k1.grad)
        arrays[f"dL_{tag}"] = npy(L.grad)
        arrays[f"dparticles_{tag}"] = npy(parts.grad)[:256]
        arrays[f"species_{tag}"] = np.asarray(species_meta(b.species))
        arrays[f"energy_{tag}"] = npy(b.energy)
        arrays[f"charges_{tag}"] = npy(b.particle_charges)
    save("grad_k1.npz", **arrays)


def gen_screen_grad():
    """Gradient of a weighted Screen.reading (cloud-in-cell) wrt quadrupole k1, particles and charges."""
    torch.manual_seed(77)
    arrays = {}
    N = 2000
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        t = lambda v: torch.tensor(v, dtype=dt)  # noqa: E731
        torch.manual_seed(77)
        beam = cheetah.ParticleBeam.from_parameters(num_particles=N, sigma_x=t(2e-4), sigma_y=t(3e-4), dtype=torch.float64).to(dt)
        k1 = torch.nn.Parameter(t(2.5))
        seg = cheetah.Segment(elements=[cheetah.Quadrupole(length=t(0.3), k1=k1, dtype=dt), cheetah.Drift(length=t(0.5), dtype=dt),
                                        cheetah.Screen(resolution=(40, 30), pixel_size=t([5e-5, 6e-5]), misalignment=t([1e-4, -5e-5]),
                                                       is_active=True, method="cloud-in-cell", name="scr", dtype=dt)])
        parts = beam.particles.clone().requires_grad_(True)
        q = beam.particle_charges.clone().requires_grad_(True)
        b = cheetah.ParticleBeam(parts, beam.energy, particle_charges=q, dtype=dt)
        seg.track(b)
        img = seg.scr.reading
        W = torch.linspace(0.5, 1.5, 40 * 30, dtype=dt).reshape(30, 40).sin()
        loss = (img * W).sum() * 1e15
        loss.backward()
        arrays[f"in_{tag}"] = npy(beam.particles)
        arrays[f"q_{tag}"] = npy(beam.particle_charges)
        arrays[f"W_{tag}"] = npy(W)
        arrays[f"loss_{tag}"] = npy(loss)
        arrays[f"img_{tag}"] = npy(img)
        arrays[f"dk1_{tag}"] = npy(k1.grad)
        arrays[f"dparticles_{tag}"] = npy(parts.grad)
        arrays[f"dq_{tag}"] = npy(q.grad)
        arrays[f"species_{tag}"] = np.asarray(species_meta(b.species))
        arrays[f"energy_{tag}"] = npy(b.energy)
    save("screen_grad.npz", **arrays)


def gen_parameter_beam():
    """Row f2: ParameterBeam tracking — the reference's own ParameterBeam consistency goldens re-exported, plus
    an active-cavity / screen-reading segment and a vectorised scan (fp64)."""
    incoming32 = load_incoming()
    incoming = incoming32.to(torch.float64).as_parameter_beam()
    arrays = {"mu": npy(incoming.mu), "cov": npy(incoming.cov), "energy": npy(incoming.energy),
              "total_charge": npy(incoming.total_charge), "species": np.asarray(species_meta(incoming.species))}
    wanted = ["Drift_ParameterBeam_linear", "Quadrupole_ParameterBeam_linear", "Dipole_ParameterBeam_linear",
              "RBend_ParameterBeam_linear", "HorizontalCorrector_ParameterBeam_default",
              "VerticalCorrector_ParameterBeam_default", "CombinedCorrector_ParameterBeam_default",
              "Cavity_ParameterBeam_default", "CustomTransferMap_ParameterBeam_identity",
              "Marker_ParameterBeam_default", "Screen_ParameterBeam_default", "Segment_ParameterBeam_default"]
    for name in wanted:
        with open(os.path.join(REF, "tests/resources/consistency_expected_outgoing", name + ".pkl"), "rb") as f:
            exp = pickle.load(f)
        arrays[f"{name}__mu"] = npy(exp.mu)
        arrays[f"{name}__cov"] = npy(exp.cov)
        arrays[f"{name}__energy"] = npy(exp.energy)
        arrays[f"{name}__s"] = npy(exp.s)
    # active cavities (SW / TW, vectorised voltage) + README segment with screen reading
    f64 = torch.float64
    t = lambda v: torch.tensor(v, dtype=f64)  # noqa: E731
    beam = cheetah.ParameterBeam.from_parameters(mu_x=t(1e-4), mu_py=t(-2e-6), sigma_tau=t(1e-4), sigma_p=t(1e-3),
                                                 cov_xpx=t(2e-10), energy=t(6e6), total_charge=t(1e-10), dtype=f64)
    arrays["b_mu"], arrays["b_cov"] = npy(beam.mu), npy(beam.cov)
    for i, (ctype, V) in enumerate((("standing_wave", 18.15975e6), ("traveling_wave", 18.15975e6),
                                    ("standing_wave", [18.15975e6, -1.0e6, 5e6]))):
        cav = cheetah.Cavity(length=t(1.0377), voltage=t(V), phase=t(30.0), frequency=t(1.3e9), cavity_type=ctype, dtype=f64)
        out = cav.track(beam)
        arrays[f"cav{i}_mu"], arrays[f"cav{i}_cov"], arrays[f"cav{i}_energy"] = npy(out.mu), npy(out.cov), npy(out.energy)
    seg = readme_segment(f64)
    seg.AREABSCR1.is_active = True
    seg.AREABSCR1.resolution = (200, 160)
    # pixels that RESOLVE the beam (sigma ~ 1.3 um ~ 6 px): the reference's pixel grid is a dtype-less
    # torch.arange (float32 positions, ~1e-11 m jitter), harmless only when the beam spans several pixels
    seg.AREABSCR1.pixel_size = t([2e-7, 3e-7])
    seg.AREABSCR1.misalignment = t([-4.4e-5, 9.5e-5])
    b2 = cheetah.ParameterBeam.from_twiss(beta_x=t(3.14), beta_y=t(42.0), energy=t(1e8), dtype=f64)
    out = seg.track(b2)
    arrays["seg_in_mu"], arrays["seg_in_cov"] = npy(b2.mu), npy(b2.cov)
    arrays["seg_out_mu"], arrays["seg_out_cov"], arrays["seg_out_s"] = npy(out.mu), npy(out.cov), npy(out.s)
    arrays["seg_reading"] = npy(seg.AREABSCR1.reading)
    for n in ("sigma_x", "sigma_y", "beta_x", "alpha_x", "emittance_x", "emittance_y"):
        arrays["seg_" + n] = npy(getattr(out, n))
    # vectorised k1 scan
    k1 = torch.linspace(-30, 30, 64, dtype=f64)
    seg2 = ares_subcell(f64, k1)
    out2 = seg2.track(b2)
    arrays["scan_k1"], arrays["scan_mu"], arrays["scan_cov"] = npy(k1), npy(out2.mu), npy(out2.cov)
    save("parameter_beam.npz", **arrays)


def gen_misc_elements():
    """Row f3: Solenoid / Undulator / Sextupole(linear) maps and the reference's consistency goldens for them."""
    f64 = {"dtype": torch.float64}
    t = lambda v: torch.tensor(v, **f64)  # noqa: E731
    elec = cheetah.Species("electron", **f64)
    arrays = {}
    i = 0
    for E in (6e6, 1e8):
        for (L, k, mx, my) in [(1.0, 1.0, 0.01, -0.02), (1.0, -2.0, 0.01, -0.02), (0.5, 0.0, 0.0, 0.0), (0.3, 7.5, 0.0, 1e-3)]:
            el = cheetah.Solenoid(length=t(L), k=t(k), misalignment=t([mx, my]), **f64)
            arrays[f"m{i}_kind"], arrays[f"m{i}_params"], arrays[f"m{i}_energy"] = np.asarray("solenoid"), np.asarray([L, k, mx, my]), np.asarray(E)
            arrays[f"m{i}_R"] = npy(el.first_order_transfer_map(t(E), elec))
            i += 1
        for (L, kx, ky, per) in [(1.0, 1.3, 0.0, 0.1), (2.0, 0.0, 0.9, 0.05), (1.0, 1.3, 0.7, 0.1), (1.0, 1.3, 0.0, 0.0), (0.5, 0.0, 0.0, 0.1)]:
            el = cheetah.Undulator(length=t(L), kx=t(kx), ky=t(ky), period=t(per), **f64)
            arrays[f"m{i}_kind"], arrays[f"m{i}_params"], arrays[f"m{i}_energy"] = np.asarray("undulator"), np.asarray([L, kx, ky, per]), np.asarray(E)
            arrays[f"m{i}_R"] = npy(el.first_order_transfer_map(t(E), elec))
            i += 1
    arrays["n_cases"] = np.asarray(i)
    keep = 512
    for name in ("Solenoid_ParticleBeam_default", "Undulator_ParticleBeam_default", "Sextupole_ParticleBeam_linear"):
        with open(os.path.join(REF, "tests/resources/consistency_expected_outgoing", name + ".pkl"), "rb") as f:
            exp = pickle.load(f)
        arrays[f"{name}__particles"] = npy(exp.particles)[..., :keep, :]
    arrays["keep"] = np.asarray(keep)
    save("misc_elements.npz", **arrays)


if __name__ == "__main__":
    which = sys.argv[1:] or ["maps", "consistency", "segment_readme", "fodo100", "k1scan", "cavity", "moments",
                             "cic", "screen", "space_charge", "grad", "screen_grad", "parameter_beam", "misc_elements"]
    for w in which:
        globals()["gen_" + w]()

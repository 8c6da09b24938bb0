#!/usr/bin/env python3
"""Fixtures for the lattice / beam import row: tests/golden/converters/.

Run in the build container only (imports /root/reference read-only through generate_golden.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_converters.py

Contents are DATA only:
  * the lattice files the reference's own converter tests read (tests/resources/bmad_tutorial_lattice.bmad, fodo.lte,
    cavity.lte, Stage4v3_9.txt: test_bmad_conversion.py, test_elegant_conversion.py, test_reading_nx_tables.py), copied
    byte for byte, plus two small files written here that exercise includes, continuation lines, wildcards, inheritance,
    reverse-Polish expressions and every element type the converters know;
  * what the reference's converters make of each of them, saved with the reference's LatticeJSON writer;
  * a synthetic Astra particle file (the reference's tests use one that is not part of its repository) and the beams the
    reference loads from it; Elegant -> Cheetah coordinate conversion and the openPMD particle-group import on seeded
    inputs.
"""
import json
import os
import shutil
import sys
import types
import warnings

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from generate_golden import REF, cheetah, np, npy, torch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "converters")

EXTRA_BMAD = """! exercises: call, continuation (& and trailing comma), wildcards, inheritance, every known type
call, file = extra_included.bmad
parameter[particle] = electron
l_cell = 2 * 0.75
q0: quadrupole, l = 0.2, k1 = k_focus
qf: q0
qd: q0, k1 = -k_focus, &
    tilt = pi / 8
quadrupole::q*[l] = 0.25
b1: sbend, l = 0.4, angle = 10 * raddeg, e1 = 0.02, e2 = 0.03, hgap = 0.015, fint = 0.4,
    fintx = 0.6, ref_tilt = 0.1
sx: sextupole, l = 0.1, k2 = 3.5 ^ 2
sol: solenoid, l = 0.3, ks = 0.9
cav: lcavity, l = 1.0, rf_frequency = 1.3e9, voltage = 2e7, phi0 = 0.05, cavity_type = traveling_wave
hk: hkicker, kick = 1e-4
vk: vkicker, kick = -2e-4
rc: rcollimator, l = 0.05, x_limit = 2e-3, y_limit = 3e-3
ec: ecollimator, x_limit = 1e-3
wig: wiggler, l = 1.2, l_period = 0.03
pa: patch
mon: monitor, l = 0.1
mon0: monitor
ins: instrument
pp: pipe, l = l_cell - sqrt(0.25)
mk: marker
odd: taylor, l = 0.07
inner: line = (qf, pp, qd)
lat: line = (mk, inner, b1, sx, sol, cav, hk, vk, rc, ec, wig, pa, mon, mon0, ins, odd)
use, lat
"""
EXTRA_BMAD_INCLUDED = "k_focus = 4.2 ! included file\n"

EXTRA_LTE = """! every Elegant type the converter knows, reverse-Polish and infix expressions
so: sole, l = 0.2
hk: hkick, l = 0.1, kick = 1e-4
vk: vkic, kick = -1e-4
kk: kicker, l = 0.1, hkick = 2e-4, vkick = 3e-4
dr: drif, l = 0.5 0.25 +
cd: csrdrift, l = 0.3
ld: lscdrif, l = 0.4
ecl: ecol, l = 0.1, x_max = 1e-3, y_max = 2e-3
rcl: rcol, x_max = 3e-3
qq: quadrupole, l = 0.2, k1 = 2.5, tilt = 0.1
sx: sextupole, l = 0.1, k2 = 10
mo: moni, l = 0.06
mo0: moni
em: ematrix, l = 0.5, c1 = 1e-3, r11 = 1, r12 = 0.5, r22 = 1, r33 = 1, r34 = 0.5, r44 = 1, r55 = 1, r66 = 1
rw: rfcw, l = 0.3, volt = 1e6, phase = 45, freq = 2.856e9
rd: rfdf, l = 0.4, voltage = 2e6, phase = 90
sb: csbend, l = 0.3, angle = 0.1, k1 = 0.2, e1 = 0.05, e2 = 0.05, tilt = 0.2, hgap = 0.01, fint = 0.4
rb: rben, l = 0.3, angle = 0.1, e1 = 0.01, e2 = 0.02, tilt = 0.3
wg: wiggler, l = 1.0, k = 1.5, poles = 20
wa: watch, filename = "w.sdds"
wk: wake
zz: twiss, l = 0.01
everything: line = (so, hk, vk, kk, dr, cd, ld, ecl, rcl, qq, sx, mo, mo0, em, rw, rd, sb, rb, wg, wa, wk, zz)
"""


def dump(segment, name):
    path = os.path.join(OUT, name + ".json")
    segment.to_lattice_json(path)
    with open(path) as f:
        print(name, len(json.load(f)["elements"]), "elements")


def main():
    os.makedirs(OUT, exist_ok=True)
    warnings.simplefilter("ignore")
    for fname in ("bmad_tutorial_lattice.bmad", "fodo.lte", "cavity.lte", "Stage4v3_9.txt"):
        shutil.copyfile(os.path.join(REF, "tests/resources", fname), os.path.join(OUT, fname))
    for fname, text in (("extra.bmad", EXTRA_BMAD), ("extra_included.bmad", EXTRA_BMAD_INCLUDED), ("extra.lte", EXTRA_LTE)):
        with open(os.path.join(OUT, fname), "w") as f:
            f.write(text)
    kw = {"dtype": torch.float64}
    dump(cheetah.Segment.from_bmad(os.path.join(OUT, "bmad_tutorial_lattice.bmad"), **kw), "bmad_tutorial_lattice")
    dump(cheetah.Segment.from_bmad(os.path.join(OUT, "extra.bmad"), sanitize_names=False, **kw), "extra_bmad")
    dump(cheetah.Segment.from_elegant(os.path.join(OUT, "fodo.lte"), "fodo", sanitize_names=False, **kw), "fodo")
    dump(cheetah.Segment.from_elegant(os.path.join(OUT, "fodo.lte"), "reversed_fodo", sanitize_names=False, **kw).flattened(),
         "reversed_fodo")
    dump(cheetah.Segment.from_elegant(os.path.join(OUT, "cavity.lte"), "cavity", **kw), "cavity")
    dump(cheetah.Segment.from_elegant(os.path.join(OUT, "extra.lte"), "everything", **kw), "extra_lte")
    dump(cheetah.Segment.from_nx_tables(os.path.join(OUT, "Stage4v3_9.txt")), "Stage4v3_9")

    arrays = {}
    # --- Astra: 300 macro-particles around a 100 MeV/c reference, 7 of them lost
    rng = np.random.default_rng(7)
    n = 300
    tab = np.zeros((n, 10))
    tab[:, 0] = rng.normal(0, 2e-4, n)
    tab[:, 1] = rng.normal(0, 3e-4, n)
    tab[:, 2] = rng.normal(0, 1e-4, n)
    tab[:, 3] = rng.normal(0, 4e2, n)
    tab[:, 4] = rng.normal(0, 5e2, n)
    tab[:, 5] = rng.normal(0, 2e5, n)
    tab[:, 6] = rng.normal(0, 1e-3, n)
    tab[:, 7] = -1e-3 / n
    tab[:, 8] = 1
    tab[:, 9] = 5
    tab[0, :7] = [0, 0, 1.25, 0, 0, 1.0e8, 4.2]
    tab[rng.choice(np.arange(1, n), 7, replace=False), 9] = -1
    np.savetxt(os.path.join(OUT, "synthetic.astra"), tab, fmt="%.12e")
    pb = cheetah.ParticleBeam.from_astra(os.path.join(OUT, "synthetic.astra"), **kw)
    qb = cheetah.ParameterBeam.from_astra(os.path.join(OUT, "synthetic.astra"), **kw)
    arrays["astra_particles"], arrays["astra_energy"] = npy(pb.particles), npy(pb.energy)
    arrays["astra_charges"] = npy(pb.particle_charges)
    arrays["astra_mu"], arrays["astra_cov"], arrays["astra_total_charge"] = npy(qb.mu), npy(qb.cov), npy(qb.total_charge)
    # --- Elegant coordinates
    from cheetah.converters.elegant import elegant_to_cheetah_coordinates
    g = torch.Generator().manual_seed(3)
    ele = torch.randn(1, 50, 6, generator=g, **kw) * torch.tensor([1e-4, 1e-3, 1e-4, 1e-3, 1e-12, 0.5], **kw)
    ele[..., 5] += 200.0
    pc = torch.tensor([199.5], **kw)   # one SDDS page (the reference's expression only broadcasts for one)
    arrays["elegant_in"], arrays["elegant_pc"] = npy(ele), npy(pc)
    arrays["elegant_out"] = npy(elegant_to_cheetah_coordinates(ele, pc))
    # --- openPMD particle group (duck-typed: the reference only reads attributes)
    m = 40
    grp = types.SimpleNamespace(
        species="electron", x=rng.normal(0, 1e-4, m), y=rng.normal(0, 1e-4, m), px=rng.normal(0, 1e3, m),
        py=rng.normal(0, 1e3, m), t=rng.normal(0, 1e-13, m), energy=5e7 + rng.normal(0, 1e4, m),
        weight=np.full(m, 1e-15), status=np.ones(m))
    for k in ("x", "y", "px", "py", "t", "energy", "weight", "status"):
        arrays["pmd_" + k] = getattr(grp, k)
    ob = cheetah.ParticleBeam.from_openpmd_particlegroup(grp, torch.tensor(5e7, **kw), **kw)
    arrays["pmd_particles"], arrays["pmd_charges"], arrays["pmd_survival"] = (
        npy(ob.particles), npy(ob.particle_charges), npy(ob.survival_probabilities))
    np.savez_compressed(os.path.join(OUT, "beams.npz"), **arrays)
    print("wrote beams.npz")


if __name__ == "__main__":
    main()

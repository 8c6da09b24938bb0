#!/usr/bin/env python3
"""Elegant SDDS particle files written BY HAND from the SDDS specification + what the reference makes of their numbers:
tests/golden/converters/elegant_bunch_{binary,ascii,colmajor_be}.sdds and tests/golden/converters/sdds_beam.npz.

Run in the build container only (imports /root/reference read-only through generate_golden.py):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_golden_sdds.py

The reference reads SDDS through the third-party `sdds` package (cheetah/converters/elegant.py:467-480), which is not in this
image and cannot be installed; cheetah_amd reads the container itself (cheetah_amd/converters/sdds_file.py). This script is the
independent side of that reader's pin: it does NOT import cheetah_amd. It
  * lays out the bytes of three files exactly as the SDDS protocol prescribes ("SDDS1" line, `!# little-endian`, namelist
    commands, `&data mode=binary`: per page an int32 row count, the parameters without a fixed value in declaration order — a
    string as int32 length + bytes —, then the rows; ASCII: one line per such parameter, the row count, one line per row;
    SDDS3 `column_major_order=1`: the table column by column; `endian=big` in the &data command) with the header elegant's
    `bunched_beam` / `watch` output carries (Step, pCentral, Charge, Particles, IDSlotsPerBunch, a fixed-value string, the
    columns x xp y yp t p dt particleID, trailing commas before `&end`, the unquoted symbol x');
  * runs the numbers it wrote through the REFERENCE's conversion — `elegant_to_cheetah_coordinates` (elegant.py:523-560) and the
    energy / charge expressions of `convert_beam` (elegant.py:497-520) — and stores what `ParticleBeam.from_elegant` must return.
One page of twelve particles: the reference's conversion broadcasts `energy - reference_energy` as (pages, N) - (pages,)
(elegant.py:563-566), which only works for a single page — the shape elegant's bunch files have.
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from generate_golden import np, torch  # noqa: E402  (puts /root/reference on the path)

from cheetah.converters.elegant import elegant_to_cheetah_coordinates  # noqa: E402
from scipy.constants import physical_constants  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "converters")
electron_mass_eV = physical_constants["electron mass energy equivalent in MeV"][0] * 1e6   # as elegant.py:17-19

HEADER = """SDDS{version}
{endian_comment}&description text="bunched-beam phase space--input: run.ele  lattice: fodo.lte", contents="bunched-beam phase space", &end
&parameter name=Step, description="Simulation step", type=long, &end
&parameter name=pCentral, symbol="p$bcen$n", units="m$be$nc", description="Reference beta*gamma", type=double, &end
&parameter name=Charge, units=C, description="Bunch charge before sampling", type=double, &end
&parameter name=Particles, description="Number of particles before sampling", type=long, &end
&parameter name=IDSlotsPerBunch, description="Number of particle ID slots reserved to a bunch", type=long, &end
&parameter name=SVNVersion, description="SVN version number", type=string, fixed_value=unknown, &end
&parameter name=Label, description="a string parameter, with a comma", type=string, &end
&column name=x, units=m, type=double,  &end
&column name=xp, symbol=x', type=double,  &end
&column name=y, units=m, type=double,  &end
&column name=yp, symbol=y', type=double,  &end
&column name=t, units=s, type=double,  &end
&column name=p, units="m$be$nc", type=double,  &end
&column name=dt, units=s, type=double,  &end
&column name=particleID, type=ulong64,  &end
{data}
"""


def pages():
    rng = np.random.default_rng(20260929)
    out = []
    for step, p0 in ((1, 195.69512),):
        n = 12
        rows = np.stack([rng.normal(size=n) * 2.1e-4, rng.normal(size=n) * 1.3e-5, rng.normal(size=n) * 1.7e-4,
                         rng.normal(size=n) * 2.2e-5, 3.3e-9 * step + rng.normal(size=n) * 1.1e-13,
                         p0 * (1 + rng.normal(size=n) * 1.5e-3), rng.normal(size=n) * 1.1e-13], axis=1)
        ids = np.arange(1, n + 1, dtype=np.uint64) + 1000 * step
        out.append({"Step": step, "pCentral": p0, "Charge": 2.5e-10 * step, "Particles": n, "IDSlotsPerBunch": 100,
                    "Label": f"page {step} of the bunch", "rows": rows, "ids": ids})
    return out


def write_binary(path, pgs, big_endian=False, column_major=False):
    e = ">" if big_endian else "<"
    data = "&data mode=binary" + (", column_major_order=1" if column_major else "") + (", endian=big" if big_endian else "") + ", &end"
    head = HEADER.format(version=3 if column_major else 1, endian_comment="" if big_endian else "!# little-endian\n", data=data)
    blob = head.encode("latin-1")
    for p in pgs:
        label = p["Label"].encode("latin-1")
        blob += struct.pack(e + "i", len(p["rows"]))
        blob += struct.pack(e + "i", p["Step"]) + struct.pack(e + "d", p["pCentral"]) + struct.pack(e + "d", p["Charge"])
        blob += struct.pack(e + "i", p["Particles"]) + struct.pack(e + "i", p["IDSlotsPerBunch"])
        blob += struct.pack(e + "i", len(label)) + label          # (SVNVersion has a fixed value: not in the pages)
        if column_major:
            for c in range(7):
                blob += b"".join(struct.pack(e + "d", float(v)) for v in p["rows"][:, c])
            blob += b"".join(struct.pack(e + "Q", int(v)) for v in p["ids"])
        else:
            for row, pid in zip(p["rows"], p["ids"]):
                blob += b"".join(struct.pack(e + "d", float(v)) for v in row) + struct.pack(e + "Q", int(pid))
    with open(path, "wb") as f:
        f.write(blob)


def write_ascii(path, pgs):
    text = HEADER.format(version=1, endian_comment="", data="&data mode=ascii, &end")
    for p in pgs:
        text += f"! page number {p['Step']}\n"
        text += f"{p['Step']}\n{p['pCentral']!r}\n{p['Charge']!r}\n{p['Particles']}\n{p['IDSlotsPerBunch']}\n\"{p['Label']}\"\n"
        text += f"         {len(p['rows'])}\n"
        for row, pid in zip(p["rows"], p["ids"]):
            text += " ".join(f"{float(v)!r}" for v in row) + f" {int(pid)}\n"
    with open(path, "w", encoding="latin-1") as f:
        f.write(text)


def main():
    pgs = pages()
    write_binary(os.path.join(OUT, "elegant_bunch_binary.sdds"), pgs)
    write_binary(os.path.join(OUT, "elegant_bunch_colmajor_be.sdds"), pgs, big_endian=True, column_major=True)
    write_ascii(os.path.join(OUT, "elegant_bunch_ascii.sdds"), pgs)
    # what the reference's convert_beam computes from `sdds.load(...)` of such a file (elegant.py:497-520), float64
    f64 = {"dtype": torch.float64}
    column_data = [[list(p["rows"][:, c]) for p in pgs] for c in range(6)]                  # columnData[:6]: [column][page][row]
    elegant_coordinates = torch.tensor(column_data, **f64).permute(1, 2, 0)
    p_central = torch.tensor([p["pCentral"] for p in pgs], **f64)
    reference_momentum_eV = p_central * electron_mass_eV
    particles = elegant_to_cheetah_coordinates(elegant_coordinates, p_central)
    energy = (reference_momentum_eV**2 + electron_mass_eV**2).sqrt()
    charges = torch.ones(particles.shape[:-1], **f64)                                       # no "q" column in elegant's output
    # without pCentral: the first particle's momentum is the reference (elegant.py:500-506)
    p_first = elegant_coordinates[..., 0, 5]
    particles_first = elegant_to_cheetah_coordinates(elegant_coordinates, p_first)
    energy_first = ((p_first * electron_mass_eV) ** 2 + electron_mass_eV**2).sqrt()
    np.savez_compressed(os.path.join(OUT, "sdds_beam.npz"), elegant_rows=np.stack([p["rows"] for p in pgs]),
                        ids=np.stack([p["ids"] for p in pgs]), p_central=p_central.numpy(), particles=particles.numpy(),
                        energy=energy.numpy(), charges=charges.numpy(), particles_first=particles_first.numpy(),
                        energy_first=energy_first.numpy(), steps=np.asarray([p["Step"] for p in pgs]),
                        bunch_charge=np.asarray([p["Charge"] for p in pgs]), labels=np.asarray([p["Label"] for p in pgs]))
    for name in ("elegant_bunch_binary.sdds", "elegant_bunch_colmajor_be.sdds", "elegant_bunch_ascii.sdds", "sdds_beam.npz"):
        print(name, os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    main()

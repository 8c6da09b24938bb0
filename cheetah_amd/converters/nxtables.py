"""NX-tables export of the ARES lattice (DESY) -> Segment (behavioural mirror of cheetah/converters/nxtables.py:9-263).

The file is a CSV with one row per installed component: NAME, CLASS (a four-letter device class) and its centre position
Z_beam. Each class maps to an element with the facility's nominal parameters, is ignored, or is a Marker; gaps between
consecutive components are filled with Drifts."""

from __future__ import annotations

import csv
from pathlib import Path

import torch

#: device classes that do not enter the beam-dynamics model
IGNORED = frozenset("RSBG MSOB MSOH MSOG VVAG BSCL MIRA BAML SCRL TEMG FCNG SOLE EOLE MSOL BELS VVAF MIRM SCRY FPSA VPUL "
                    "SOLC SCRE SOLX ICTB BSCS".split())
#: device classes kept as position markers
MARKERS = frozenset("SOLG BCMG EOLG SOLS EOLS SOLA EOLA SOLT BSTB TORF EOLT SOLO EOLO SOLB EOLB ECHA MKBB MKBE MKPM EOLC "
                    "SOLM EOLM SOLH BSCD STDE ECHS EOLH WINA LINA EOLX".split())
#: screen classes: (resolution, pixel size)
SCREENS = {
    "BSCX": ((2464, 2056), (0.00343e-3, 0.00247e-3)),
    "BSCR": ((2448, 2040), (3.5488e-6, 2.5003e-6)),
    "BSCM": ((2448, 2040), (3.5488e-6, 2.5003e-6)),
    "BSCO": ((2448, 2040), (3.5488e-6, 2.5003e-6)),
    "BSCA": ((2448, 2040), (3.5488e-6, 2.5003e-6)),
    "BSCE": ((2464, 2056), (0.00998e-3, 0.00715e-3)),
    "SCRD": ((2464, 2056), (0.00998e-3, 0.00715e-3)),
}


def _acc():
    from .. import accelerator
    return accelerator


def translate_element(row: list[str], header: list[str]) -> dict | None:
    """One table row -> {"element": Element, "s_position": float}, or None for an ignored class."""
    a = _acc()
    t = torch.tensor
    device_class, name = row[header.index("CLASS")], row[header.index("NAME")]
    s_position = float(row[header.index("Z_beam")])
    if device_class in IGNORED:
        return None
    if device_class in MARKERS:
        element = a.Marker(name=name)
    elif device_class in SCREENS:
        resolution, pixel_size = SCREENS[device_class]
        element = a.Screen(name=name, resolution=resolution, pixel_size=t(pixel_size), binning=1)
    elif device_class == "MCXG":   # combined steerer: an H and a V coil at the same place
        assert name[6] == "X"
        element = a.Segment(elements=[a.HorizontalCorrector(name=name[:6] + "H" + name[7:], length=t(5e-05)),
                                      a.VerticalCorrector(name=name[:6] + "V" + name[7:], length=t(5e-05))], name=name)
    elif device_class in ("BPMG", "BPML"):
        element = a.BPM(name=name)
    elif device_class in ("SLHG", "SLHB", "SLHS"):
        element = a.Aperture(name=name, x_max=t(float("inf")), y_max=t(float("inf")),
                             shape="elliptical" if device_class == "SLHG" else "rectangular")
    elif device_class == "MCHM":
        element = a.HorizontalCorrector(name=name, length=t(0.02))
    elif device_class == "MCVM":
        element = a.VerticalCorrector(name=name, length=t(0.02))
    elif device_class == "MBHL":
        element = a.Dipole(name=name, length=t(0.322))
    elif device_class == "MBHB":
        element = a.Dipole(name=name, length=t(0.22))
    elif device_class == "MBHO":
        element = a.Dipole(name=name, length=t(0.43852543421396856), angle=t(0.8203047484373349),
                           dipole_e2=t(-0.7504915783575616))
    elif device_class == "MQZM":
        element = a.Quadrupole(name=name, length=t(0.122))
    elif device_class == "RSBL":
        element = a.Cavity(name=name, length=t(4.139), frequency=t(2.998e9), voltage=t(76e6))
    elif device_class == "RXBD":
        element = a.Cavity(name=name, length=t(1.0), frequency=t(11.9952e9), voltage=t(0.0))
    elif device_class == "UNDA":
        element = a.Undulator(name=name, length=t(0.25))
    else:
        raise ValueError(f"Encountered unknown class {device_class} for element {name}")
    return {"element": element, "s_position": s_position}


def convert_lattice(filepath: Path):
    """NX-tables file -> flattened Segment named after the file, components sorted by position, Drifts in between."""
    a = _acc()
    filepath = Path(filepath)
    with open(filepath, "r") as f:
        rows = list(csv.reader(f, delimiter=","))
    header, rows = rows[0], rows[1:]
    placed = sorted((item for item in (translate_element(row, header) for row in rows) if item is not None),
                    key=lambda item: item["s_position"])
    elements = [placed[0]["element"]]
    for previous, current in zip(placed[:-1], placed[1:]):
        gap = (current["s_position"] - previous["s_position"]
               - previous["element"].length / 2 - current["element"].length / 2)
        assert gap >= 0.0, f"Elements {previous['element'].name} and {current['element'].name} overlap by {gap}."
        if gap > 0.0:
            elements.append(a.Drift(name=f"DRIFT_{previous['element'].name}_{current['element'].name}",
                                    length=torch.as_tensor([float(gap)])))
        elements.append(current["element"])
    return a.Segment(elements=elements, name=filepath.stem).flattened()

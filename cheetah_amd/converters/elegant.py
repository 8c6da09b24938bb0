"""Elegant lattice files (.lte) -> Segment, Elegant coordinates -> Cheetah coordinates (behavioural mirror of
cheetah/converters/elegant.py:20-567). One `RULES` row per group of Elegant type names; defaults and silently accepted
properties are the reference's."""

from __future__ import annotations

import math
import warnings
from pathlib import Path

import torch

from ..warnings import NoBeamPropertiesInLatticeWarning, UnknownElementWarning
from . import lattice_text

SHARED = ["element_type", "group"]
ELECTRON_MASS_EV = 0.51099895069e6   # scipy.constants (CODATA 2022), as the reference
SPEED_OF_LIGHT = 299792458.0


def _acc():
    from .. import accelerator
    return accelerator


def _drift(p, n, sn, t):
    return _acc().Drift(length=t(p.get("l", 0.0)), name=n, sanitize_name=sn)


def _collimator(shape):
    def build(p, n, sn, t):
        a = _acc()
        return a.Segment(elements=[
            a.Drift(length=t(p.get("l", 0.0)), name=n + "_drift", sanitize_name=sn),
            a.Aperture(x_max=t(p.get("x_max", math.inf)), y_max=t(p.get("y_max", math.inf)), shape=shape,
                       name=n + "_aperture", sanitize_name=sn)], name=n + "_segment", sanitize_name=sn)
    return build


def _monitor(p, n, sn, t):
    a = _acc()
    if "l" not in p:
        return a.BPM(name=n, sanitize_name=sn)
    half = p.get("l", 0.0) / 2
    return a.Segment(elements=[a.Drift(length=t(half), name=n + "_predrift", sanitize_name=sn),
                               a.BPM(name=n, sanitize_name=sn),
                               a.Drift(length=t(half), name=n + "_postdrift", sanitize_name=sn)],
                     name=n + "_segment", sanitize_name=sn)


def _ematrix(p, n, sn, t):
    if p.get("order", 1) != 1:
        raise ValueError("Only first order modelling is supported")
    R = t([[p.get(f"r{i + 1}{j + 1}", 0.0) for j in range(6)] + [p.get(f"c{i + 1}", 0.0)] for i in range(6)]
          + [[0.0] * 6 + [1.0]])      # Elegant's R starts from zero, C is the constant term
    return _acc().CustomTransferMap(length=t(p.get("l", 0.0)), predefined_transfer_map=R, name=n, sanitize_name=sn)


def _cavity(p, n, sn, t):
    # Elegant: maximum acceleration at 90 deg; here at 0 deg
    return _acc().Cavity(length=t(p.get("l", 0.0)), phase=t(p.get("phase", 0.0) - 90), voltage=t(p.get("volt", 0.0)),
                         frequency=t(p.get("freq", 500e6)), name=n, sanitize_name=sn)


def _wiggler(p, n, sn, t):
    length = p.get("l", 0.0)
    period = 2.0 * length / p["poles"] if "poles" in p else 0.0   # two poles per period
    return _acc().Undulator(length=t(length), period=t(period), kx=t(p.get("k", 0.0)), name=n, sanitize_name=sn)


def _beam_info(p, n, sn, t):
    warnings.warn(f"Information provided in element {n} of type {p['element_type']} cannot be imported automatically. "
                  "Consider manually providing the correct information.", category=NoBeamPropertiesInLatticeWarning,
                  stacklevel=4)
    return _acc().Marker(name=n, sanitize_name=sn)


RULES = [
    # (type names, extra understood properties (None: unchecked), builder)                       elegant.py lines
    (("sole",), ["l"], lambda p, n, sn, t: _acc().Solenoid(length=t(p.get("l", 0.0)), name=n, sanitize_name=sn)),  # 61-68
    (("hkick", "hkic"), ["l", "kick"], lambda p, n, sn, t: _acc().HorizontalCorrector(                      # 69-76
        length=t(p.get("l", 0.0)), angle=t(p.get("kick", 0.0)), name=n, sanitize_name=sn)),
    (("vkick", "vkic"), ["l", "kick"], lambda p, n, sn, t: _acc().VerticalCorrector(                        # 77-84
        length=t(p.get("l", 0.0)), angle=t(p.get("kick", 0.0)), name=n, sanitize_name=sn)),
    (("kick", "kicker"), ["l", "hkick", "vkick"], lambda p, n, sn, t: _acc().CombinedCorrector(             # 85-98
        length=t(p.get("l", 0.0)), horizontal_angle=t(p.get("hkick", 0.0)), vertical_angle=t(p.get("vkick", 0.0)),
        name=n, sanitize_name=sn)),
    (("mark", "marker"), [], lambda p, n, sn, t: _acc().Marker(name=n, sanitize_name=sn)),                  # 99-103
    (("drift", "drif", "csrdrift", "csrdrif", "lscdrift", "lscdrif"), ["l"], _drift),                      # 104-127
    (("ecol",), ["l", "x_max", "y_max"], _collimator("elliptical")),                                        # 128-155
    (("rcol",), ["l", "x_max", "y_max"], _collimator("rectangular")),                                       # 156-183
    (("quad", "quadrupole", "kquad"), ["l", "k1", "tilt"], lambda p, n, sn, t: _acc().Quadrupole(           # 184-195
        length=t(p.get("l", 0.0)), k1=t(p.get("k1", 0.0)), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    (("sext", "sextupole"), ["l", "k2", "tilt"], lambda p, n, sn, t: _acc().Sextupole(                      # 196-207
        length=t(p.get("l", 0.0)), k2=t(p.get("k2", 0.0)), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    (("moni",), ["l"], _monitor),                                                                           # 208-235
    (("ematrix",), ["l", "order", "c[1-6]", "r[1-6][1-6]"], _ematrix),                                      # 236-266
    (("rfca", "rfcw"), ["l", "phase", "volt", "freq"], _cavity),                                            # 267-294
    (("rfdf",), ["l", "phase", "voltage", "freq"], lambda p, n, sn, t: _acc().TransverseDeflectingCavity(   # 295-308
        length=t(p.get("l", 0.0)), phase=t(p.get("phase", 0.0) - 90), voltage=t(p.get("voltage", 0.0)),
        frequency=t(p.get("freq", 2.856e9)), name=n, sanitize_name=sn)),
    (("sben", "csbend"), ["l", "angle", "k1", "e1", "e2", "tilt", "hgap", "fint"], lambda p, n, sn, t: _acc().Dipole(  # 309-326
        length=t(p.get("l", 0.0)), angle=t(p.get("angle", 0.0)), k1=t(p.get("k1", 0.0)), dipole_e1=t(p.get("e1", 0.0)),
        dipole_e2=t(p.get("e2", 0.0)), tilt=t(p.get("tilt", 0.0)), gap=t(2.0 * p.get("hgap", 0.0)),
        fringe_integral=t(p.get("fint", 0.5)), name=n, sanitize_name=sn)),
    (("rben",), ["l", "angle", "e1", "e2", "tilt"], lambda p, n, sn, t: _acc().RBend(                        # 327-340
        length=t(p.get("l", 0.0)), angle=t(p.get("angle", 0.0)), rbend_e1=t(p.get("e1", 0.0)),
        rbend_e2=t(p.get("e2", 0.0)), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    (("csrcsben", "csrcsbend"), ["l", "angle", "k1", "e1", "e2", "tilt"], lambda p, n, sn, t: _acc().Dipole(  # 341-355
        length=t(p.get("l", 0.0)), angle=t(p.get("angle", 0.0)), k1=t(p.get("k1", 0.0)), dipole_e1=t(p.get("e1", 0.0)),
        dipole_e2=t(p.get("e2", 0.0)), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    (("wiggler",), ["l", "k", "poles"], _wiggler),                                                          # 356-372
    (("watch",), ["filename"], lambda p, n, sn, t: _acc().Marker(name=n, sanitize_name=sn)),                # 373-377
    (("charge", "wake"), None, _beam_info),                                                                 # 378-388
]
_BY_TYPE = {name: (understood, build) for names, understood, build in RULES for name in names}


def convert_element(name: str, context: dict, sanitize_name=None, device=None, dtype=None):
    """One entry of a parsed Elegant context -> Element; `-name` is the reversed beam line `name`."""
    kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
    tensor = lambda v: torch.tensor(v, **kw)  # noqa: E731
    is_reversed = name.startswith("-")
    name = name.removeprefix("-")
    parsed = context[name]
    if isinstance(parsed, list):
        segment = _acc().Segment(
            elements=[convert_element(member, context, sanitize_name, device, dtype) for member in parsed], name=name,
            sanitize_name=sanitize_name)
        return segment.reversed() if is_reversed else segment
    if not (isinstance(parsed, dict) and "element_type" in parsed):
        raise ValueError(f"Unknown Elegant element type for {name = }")  # noqa: E202, E251
    rule = _BY_TYPE.get(parsed["element_type"])
    if rule is None:
        warnings.warn(f"Element {name} of type {parsed['element_type']} cannot be converted correctly. Using drift section "
                      "instead.", category=UnknownElementWarning, stacklevel=2)
        return _drift(parsed, name, sanitize_name, tensor)
    understood, build = rule
    if understood is not None:
        lattice_text.check_understood(SHARED + understood, parsed)
    return build(parsed, name, sanitize_name, tensor)


def convert_lattice(elegant_lattice_file_path: Path, name: str, sanitize_names=None, device=None, dtype=None):
    """Elegant lattice file -> the beam line `name` as a Segment (elegant.py:409-451)."""
    context = lattice_text.parse(Path(elegant_lattice_file_path))
    return convert_element(name, context, sanitize_names, device, dtype)


def elegant_to_cheetah_coordinates(elegant_coordinates: torch.Tensor, p_central: torch.Tensor) -> torch.Tensor:
    """(…, N, 6) Elegant rows [x, x', y, y', t, p = beta gamma] -> (…, N, 7) Cheetah rows (elegant.py:523-567)."""
    p0 = p_central.unsqueeze(-1)
    x, xp, y, yp, time, p = elegant_coordinates.unbind(dim=-1)
    ref_momentum_eV = p0 * ELECTRON_MASS_EV
    ref_energy_eV = (ref_momentum_eV**2 + ELECTRON_MASS_EV**2).sqrt()
    energy_eV = ((p * ELECTRON_MASS_EV) ** 2 + ELECTRON_MASS_EV**2).sqrt()
    rel = 1.0 + (p - p0) / p0                                   # P / p0
    slope_norm = (1.0 + xp.square() + yp.square()).sqrt()
    return torch.stack([x, xp * rel / slope_norm, y, yp * rel / slope_norm, time * SPEED_OF_LIGHT,
                        (energy_eV - ref_energy_eV) / ref_momentum_eV, torch.ones_like(x)], dim=-1)


def convert_beam(file_path: Path, device=None, dtype=None):
    """Elegant SDDS particle file -> (particles, reference energy, charges) (elegant.py:454-520). The file is read by the
    `sdds` package when it is installed (as the reference does) and by `converters.sdds_file` otherwise."""
    kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
    try:
        import sdds
    except ImportError:
        from . import sdds_file

        data = sdds_file.load(file_path)
    else:
        data = sdds.load(str(file_path))
    columns = list(data.columnName[:6])
    if columns == ["r", "pz", "pr", "pphi", "t", "q"]:
        raise ValueError("The beam distribution is stored in the spiffe format, which is not currently supported. Use "
                         "spiffe2elegant to convert the beam first.")
    if columns != ["x", "xp", "y", "yp", "t", "p"]:
        raise ValueError("The first six columns of the SDDS file do not match the expected Elegant beam convention. "
                         "Please ensure the SDDS file is in the correct format.")
    coords = torch.tensor(data.columnData[:6], **kw).permute(1, 2, 0)         # (pages, particles, 6)
    p_central = (torch.tensor(data.getParameterValueList("pCentral"), **kw) if "pCentral" in data.parameterName
                 else coords[..., 0, 5])
    particles = elegant_to_cheetah_coordinates(coords, p_central)
    energy = ((p_central * ELECTRON_MASS_EV) ** 2 + ELECTRON_MASS_EV**2).sqrt()
    charges = (torch.tensor(data.getColumnValueLists("q"), **kw) if "q" in data.columnName
               else torch.ones(particles.shape[:-1], **kw))
    return particles, energy, charges

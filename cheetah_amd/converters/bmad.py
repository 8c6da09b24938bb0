"""Bmad lattice files -> Segment (behavioural mirror of cheetah/converters/bmad.py:17-349).

The reference walks an if-chain per element type; here each Bmad type is one row of `RULES`:
(understood properties, builder). Defaults, the properties each type accepts silently and the warnings are the
reference's (bmad.py line numbers next to every row)."""

from __future__ import annotations

import math
import os
import warnings
from pathlib import Path

import torch

from ..warnings import UnknownElementWarning
from . import lattice_text

SHARED = ["element_type", "alias", "type"]


def _acc():
    from .. import accelerator
    return accelerator


def _drift_or_marker(p, name, sn, t):
    a = _acc()
    return a.Drift(length=t(p["l"]), name=name, sanitize_name=sn) if "l" in p else a.Marker(name=name, sanitize_name=sn)


def _collimator(shape):
    def build(p, name, sn, t):
        a = _acc()
        return a.Segment(elements=[
            a.Drift(length=t(p.get("l", 0.0)), name=name + "_drift", sanitize_name=sn),
            a.Aperture(x_max=t(p.get("x_limit", math.inf)), y_max=t(p.get("y_limit", math.inf)), shape=shape,
                       name=name + "_aperture", sanitize_name=sn)], name=name, sanitize_name=sn)
    return build


RULES = {
    # type: (extra understood properties, builder(properties, name, sanitize_name, tensor factory))
    "marker": ([], lambda p, n, sn, t: _acc().Marker(name=n, sanitize_name=sn)),                         # :56-58
    "monitor": (["l"], _drift_or_marker),                                                                  # :59-68
    "instrument": (["l"], _drift_or_marker),                                                               # :69-78
    "pipe": (["l", "descrip"], lambda p, n, sn, t: _acc().Drift(length=t(p["l"]), name=n, sanitize_name=sn)),   # :79-87
    "drift": (["l", "descrip"], lambda p, n, sn, t: _acc().Drift(length=t(p["l"]), name=n, sanitize_name=sn)),  # :88-96
    "hkicker": (["kick"], lambda p, n, sn, t: _acc().HorizontalCorrector(                                  # :97-104
        length=t(p.get("l", 0.0)), angle=t(p.get("kick", 0.0)), name=n, sanitize_name=sn)),
    "vkicker": (["kick"], lambda p, n, sn, t: _acc().VerticalCorrector(                                    # :105-112
        length=t(p.get("l", 0.0)), angle=t(p.get("kick", 0.0)), name=n, sanitize_name=sn)),
    "sbend": (["hgap", "l", "angle", "e1", "e2", "fint", "fintx", "ref_tilt"],                             # :113-137
              lambda p, n, sn, t: _acc().Dipole(
                  length=t(p["l"]), gap=t(2 * p.get("hgap", 0.0)), angle=t(p.get("angle", 0.0)), dipole_e1=t(p["e1"]),
                  dipole_e2=t(p.get("e2", 0.0)), tilt=t(p.get("ref_tilt", 0.0)), fringe_integral=t(p.get("fint", 0.0)),
                  fringe_integral_exit=t(p["fintx"]) if "fintx" in p else None, name=n, sanitize_name=sn)),
    "quadrupole": (["l", "k1", "tilt"], lambda p, n, sn, t: _acc().Quadrupole(                             # :138-148
        length=t(p["l"]), k1=t(p["k1"]), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    "sextupole": (["l", "k2", "tilt"], lambda p, n, sn, t: _acc().Sextupole(                               # :149-159
        length=t(p["l"]), k2=t(p["k2"]), tilt=t(p.get("tilt", 0.0)), name=n, sanitize_name=sn)),
    "solenoid": (["l", "ks"], lambda p, n, sn, t: _acc().Solenoid(                                         # :160-167
        length=t(p["l"]), k=t(p["ks"]), name=n, sanitize_name=sn)),
    "lcavity": (["l", "rf_frequency", "voltage", "phi0"], lambda p, n, sn, t: _acc().Cavity(               # :168-186
        length=t(p["l"]), voltage=t(p.get("voltage", 0.0)),
        phase=-(t(p.get("phi0", 0.0)) * 2 * math.pi).rad2deg(),   # Bmad: phase in units of 2 pi, opposite sign
        frequency=t(p["rf_frequency"]), cavity_type=p["cavity_type"], name=n, sanitize_name=sn)),
    "rcollimator": (["l", "x_limit", "y_limit"], _collimator("rectangular")),                              # :187-217
    "ecollimator": (["l", "x_limit", "y_limit"], _collimator("elliptical")),                               # :218-248
    "wiggler": (["l", "l_period"], lambda p, n, sn, t: _acc().Undulator(                                   # :249-261
        length=t(p["l"]), period=t(p["l_period"]), name=n, sanitize_name=sn)),
    "patch": (["l"], lambda p, n, sn, t: _acc().Drift(length=t(p.get("l", 0.0)), name=n, sanitize_name=sn)),   # :262-269
}


def convert_element(name: str, context: dict, sanitize_name=None, device=None, dtype=None):
    """One entry of a parsed Bmad context -> Element (lines -> Segment, recursively)."""
    kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
    tensor = lambda v: torch.tensor(v, **kw)  # noqa: E731
    parsed = context[name]
    if isinstance(parsed, list):
        return _acc().Segment(elements=[convert_element(member, context, sanitize_name, device, dtype) for member in parsed],
                              name=name, sanitize_name=sanitize_name)
    if not (isinstance(parsed, dict) and "element_type" in parsed):
        raise ValueError(f"Unknown Bmad element type for {name = }")  # noqa: E202, E251
    rule = RULES.get(parsed["element_type"])
    if rule is None:
        warnings.warn(f"Element {name} of type {parsed['element_type']} cannot be converted correctly. Using drift section "
                      "instead.", category=UnknownElementWarning, stacklevel=2)
        return _acc().Drift(length=tensor(parsed.get("l", 0.0)), name=name, sanitize_name=sanitize_name)
    understood, build = rule
    lattice_text.check_understood(SHARED + understood, parsed)
    return build(parsed, name, sanitize_name, tensor)


def convert_lattice(bmad_lattice_file_path: Path, environment_variables: dict | None = None, sanitize_names=None,
                    device=None, dtype=None):
    """Bmad lattice file -> Segment of the line named by its `use` statement (bmad.py:283-349)."""
    if environment_variables is not None:
        os.environ.update(environment_variables)
    path = Path(bmad_lattice_file_path)
    path = Path(*[os.environ[part[1:]] if part.startswith("$") else part for part in path.parts])
    context = lattice_text.parse(path)
    return convert_element(context["__use__"], context, sanitize_names, device, dtype)

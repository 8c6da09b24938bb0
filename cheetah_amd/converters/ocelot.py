"""Ocelot cells -> Segment (behavioural mirror of cheetah/converters/ocelot.py:9-219).

The reference dispatches with `isinstance` on Ocelot's classes, which needs Ocelot installed. Ocelot objects are plain
attribute bags, so this converter is duck-typed: it looks at the class names along the object's MRO, in the reference's
order of checks (so, as there, every subclass of `Bend` — SBend and RBend included — becomes a Dipole). No import of
Ocelot is needed; any object with the same class names and attributes converts."""

from __future__ import annotations

import warnings

import torch

from ..warnings import DefaultParameterWarning, UnknownElementWarning


def _acc():
    from .. import accelerator
    return accelerator


def _is(element, class_name: str) -> bool:
    return any(klass.__name__ == class_name for klass in type(element).__mro__)


def convert_element(element, sanitize_name=None, device=None, dtype=None):
    kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
    t = lambda v: torch.tensor(v, **kw)  # noqa: E731
    a = _acc()
    common = {"name": element.id, "sanitize_name": sanitize_name}
    if _is(element, "Drift"):
        return a.Drift(length=t(element.l), **common)
    if _is(element, "Quadrupole"):
        return a.Quadrupole(length=t(element.l), k1=t(element.k1), **common)
    if _is(element, "Sextupole"):
        return a.Sextupole(length=t(element.l), k2=t(element.k2), **common)
    if _is(element, "Solenoid"):
        return a.Solenoid(length=t(element.l), k=t(element.k), **common)
    if _is(element, "Hcor"):
        return a.HorizontalCorrector(length=t(element.l), angle=t(element.angle), **common)
    if _is(element, "Vcor"):
        return a.VerticalCorrector(length=t(element.l), angle=t(element.angle), **common)
    if _is(element, "Bend") or _is(element, "SBend"):
        return a.Dipole(length=t(element.l), angle=t(element.angle), dipole_e1=t(element.e1), dipole_e2=t(element.e2),
                        tilt=t(element.tilt), fringe_integral=t(element.fint), fringe_integral_exit=t(element.fintx),
                        gap=t(element.gap), **common)
    if _is(element, "RBend"):  # only reached for an RBend that does not derive from Bend
        return a.RBend(length=t(element.l), angle=t(element.angle), rbend_e1=t(element.e1) - element.angle / 2,
                       rbend_e2=t(element.e2) - element.angle / 2, tilt=t(element.tilt), fringe_integral=t(element.fint),
                       fringe_integral_exit=t(element.fintx), gap=t(element.gap), **common)
    for class_name, extra in (("Cavity", {"cavity_type": "standing_wave"}), ("TWCavity", {"cavity_type": "traveling_wave"}),
                              ("TDCavity", {})):
        if _is(element, class_name):   # Ocelot voltages are in GV
            return a.Cavity(length=t(element.l), voltage=t(element.v) * 1e9, frequency=t(element.freq), phase=t(element.phi),
                            **extra, **common)
    if _is(element, "Monitor") and "BSC" in element.id:
        warnings.warn("Diagnostic screen was converted with default screen properties.", category=DefaultParameterWarning,
                      stacklevel=2)
        return a.Screen(resolution=(2448, 2040), pixel_size=t([3.5488e-6, 2.5003e-6]), **common)
    if _is(element, "Monitor") and "BPM" in element.id:
        return a.BPM(**common)
    if _is(element, "Marker") or _is(element, "Monitor"):
        return a.Marker(**common)
    if _is(element, "Undulator"):
        return a.Undulator(length=t(element.l), period=t(element.lperiod), kx=t(element.Kx), ky=t(element.Ky), **common)
    if _is(element, "Aperture"):
        return a.Aperture(x_max=t(element.xmax), y_max=t(element.ymax),
                          shape={"rect": "rectangular", "elip": "elliptical"}[element.type], is_active=True, **common)
    warnings.warn(f"Unknown element {element.id} of type {type(element)}, replacing with drift section.",
                  category=UnknownElementWarning, stacklevel=2)
    return a.Drift(length=t(element.l), **common)


def subcell_of_ocelot(cell: list, start: str, end: str) -> list:
    """The elements of `cell` from the one with id `start` up to and including the one with id `end`."""
    ids = [el.id for el in cell]
    if start not in ids:
        return []
    first = ids.index(start)
    last = ids.index(end, first) if end in ids[first:] else len(cell) - 1
    return list(cell[first:last + 1])

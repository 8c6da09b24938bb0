"""LatticeJSON import / export (file format of cheetah/latticejson.py:1-260, the
https://github.com/nobeam/latticejson convention):

    {"version": ..., "title": ..., "info": ..., "root": <lattice name>,
     "elements": {name: [class name, {feature: value, ...}]},
     "lattices": {lattice name: [element or lattice names, in order]}}

Every defining feature of an element is written (tensors as nested lists, sub-elements by name with their own
entry, `metadata` verbatim), so files written by the reference load here and vice versa.
"""

from __future__ import annotations

import json
from typing import Any

import torch


def _plain(value: Any) -> Any:
    """Tensor / Parameter -> nested Python lists; everything else unchanged (latticejson.py:9-23)."""
    return value.tolist() if isinstance(value, torch.Tensor) else value


def convert_element(element, elements_dict: dict | None = None) -> tuple[str, str, dict]:
    """(name, class name, parameters) of an element; element-valued features are stored by name and added to
    `elements_dict` (latticejson.py:26-59)."""
    from .accelerator import Element

    if elements_dict is None:
        elements_dict = {}
    params = {}
    for feature in element.defining_features:
        if feature == "name":
            continue
        value = getattr(element, feature)
        if isinstance(value, Element):
            sub_name, sub_class, sub_params = convert_element(value, elements_dict)
            elements_dict[sub_name] = [sub_class, sub_params]
            params[feature] = sub_name
        else:
            params[feature] = _plain(value)
    params["metadata"] = element.metadata  # not a defining feature: it does not affect the simulation
    return element.name, element.__class__.__name__, params


def convert_segment(segment) -> tuple[dict, dict]:
    """(elements, lattices) dictionaries of a segment and its nested segments (latticejson.py:62-92)."""
    from .accelerator import Segment

    elements, lattices, cell = {}, {}, []
    for element in segment.elements:
        if isinstance(element, Segment):
            sub_elements, sub_lattices = convert_segment(element)
            elements.update(sub_elements)
            lattices.update(sub_lattices)
        else:
            _, cls_name, params = convert_element(element, elements)
            elements[element.name] = [cls_name, params]
        cell.append(element.name)
    lattices[segment.name] = cell
    return elements, lattices


class CompactJSONEncoder(json.JSONEncoder):
    """Indent the first two levels only, one element per line (the latticejson project's formatting)."""

    def encode(self, obj, level=0):
        if isinstance(obj, dict) and level < 2:
            pad = (level + 1) * self.indent * " "
            body = ",\n".join(f"{pad}{json.dumps(k)}: {self.encode(v, level=level + 1)}" for k, v in obj.items())
            return f"{{\n{body}\n{level * self.indent * ' '}}}" + ("\n" if level == 0 else "")
        return json.dumps(obj)


def save_cheetah_model(segment, filename: str, title: str | None = None,
                       info: str = "This is a placeholder lattice description") -> None:
    """latticejson.py:95-130"""
    if title is None:
        title = segment.name if segment.name is not None else "Unnamed Lattice"
    lattice_dict = {"version": "cheetah-0.8", "title": title, "info": info,
                    "root": segment.name if segment.name is not None else "cell"}
    lattice_dict["elements"], lattice_dict["lattices"] = convert_segment(segment)
    with open(filename, "w") as f:
        f.write(json.dumps(lattice_dict, cls=CompactJSONEncoder, indent=4))


def _feature(value: Any, device=None, dtype=None) -> Any:
    """Numbers (and lists containing a float) become tensors; str / bool / int (and lists of only those), dicts and
    None stay Python values, as every element constructor expects (latticejson.py:156-180)."""
    keep = (str, bool, int)
    if value is None or isinstance(value, keep) or isinstance(value, dict):
        return value
    if isinstance(value, (tuple, list)) and all(isinstance(v, keep) for v in value):
        return value
    return torch.tensor(value, device=device, dtype=dtype)


def parse_element(name: str, lattice_dict: dict, device=None, dtype=None):
    from . import accelerator

    cls_name, params = lattice_dict["elements"][name]
    cls = getattr(accelerator, cls_name)
    converted = {
        key: (parse_element(value, lattice_dict, device=device, dtype=dtype)
              if isinstance(value, str) and value in lattice_dict["elements"] else _feature(value, device, dtype))
        for key, value in params.items()
    }
    # unlike the reference (latticejson.py:207) the factory kwargs are forwarded, so defaulted tensors of files that
    # omit a feature land on the requested device as well
    return cls(name=name, **converted, device=device, dtype=dtype)


def parse_segment(name: str, lattice_dict: dict, device=None, dtype=None):
    from .accelerator import Segment

    elements = [
        (parse_segment if element_name in lattice_dict["lattices"] else parse_element)(element_name, lattice_dict,
                                                                                      device=device, dtype=dtype)
        for element_name in lattice_dict["lattices"][name]
    ]
    return Segment(elements=elements, name=name)


def load_cheetah_model(filename: str, device=None, dtype=None):
    """latticejson.py:241-260"""
    dtype = dtype if dtype is not None else torch.get_default_dtype()
    with open(filename, "r") as f:
        lattice_dict = json.load(f)
    return parse_segment(lattice_dict["root"], lattice_dict, device=device, dtype=dtype)

"""Segment (mirror of cheetah/accelerator/segment.py:45-71, 525-574, 658-700 for the tracking path).

`track` follows the reference's orchestration: an all-skippable lattice is composed into ONE 7x7 map
(`chx_compose_maps`, fp64 accumulation) and applied in one pass over the particles; otherwise the
element list is partitioned into maximal skippable runs (each composed and applied once) and the
non-skippable elements (active Cavity / Screen / SpaceChargeKick / BPM / Aperture) are tracked one by
one. Unlike the reference no nn.Module sub-segments are built per call, and the composed map of a run
is cached against the (revision, tensor version) token of its elements.

`track_elementwise` is the merge-free variant (`for e in elements: beam = e.track(beam)`): the E maps
are applied back to back by `chx_track_elementwise` / `chx_track_fused` from a single C call.
"""

from __future__ import annotations

from copy import deepcopy

import torch
from torch import nn

from .. import _ops
from ..particles.particle_beam import ParticleBeam
from ..particles.species import Species
from .element import Element


class Segment(Element):
    """Ordered sequence of elements."""

    supported_tracking_methods = ["linear"]

    def __init__(self, elements: list[Element], name=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None) -> None:
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)
        del self._buffers["length"]  # `length` is a derived property here (segment.py:54-58)
        self.elements = nn.ModuleList(elements)
        by_name: dict[str, list[Element]] = {}
        for e in elements:
            by_name.setdefault(e.name, []).append(e)
        self.__dict__["_by_name"] = by_name
        self.__dict__["_run_cache"] = {}

    def __getattr__(self, name: str):
        by_name = self.__dict__.get("_by_name")
        if by_name is not None and name in by_name:
            found = by_name[name]
            return found[0] if len(found) == 1 else found
        return super().__getattr__(name)

    @property
    def is_skippable(self) -> bool:
        return all(e.is_skippable for e in self.elements)

    @property
    def length(self) -> torch.Tensor:
        total = None
        for e in self.elements:
            total = e.length if total is None else total + e.length
        return total

    # ---- composition ---------------------------------------------------------------------------------
    @staticmethod
    def _run_token(elements, energy, species):
        tok = [id(energy), energy._version, species.mass_eV_float, species.num_elementary_charges_float]
        for e in elements:
            tok.append(e.__dict__["_revision"])
            for t in e._buffers.values():
                if t is not None:
                    tok.append(t._version)
            for t in e._parameters.values():
                if t is not None:
                    tok.append(t._version)
                    tok.append(t.requires_grad)
            if isinstance(e, Segment):
                tok.append(Segment._run_token(list(e.elements), energy, species))
        return tuple(tok)

    def _compose_run(self, key, elements, energy: torch.Tensor, species: Species) -> torch.Tensor:
        """Composed map of a run of skippable elements, cached per run."""
        cacheable = not (energy.requires_grad or species.mass_eV.requires_grad)
        token = self._run_token(elements, energy, species) if cacheable else None
        cache = self.__dict__["_run_cache"]
        hit = cache.get(key)
        if cacheable and hit is not None and hit[0] == token and not hit[1].requires_grad:
            return hit[1]
        maps = [e.first_order_transfer_map(energy, species) for e in elements]
        dtype, device = maps[0].dtype, maps[0].device
        batch_shape = torch.broadcast_shapes(energy.shape, *[m.shape[:-2] for m in maps])
        tm = _ops.compose_maps(maps, batch_shape, dtype, device)
        if cacheable:
            cache[key] = (token, tm, energy)  # `energy` kept alive so its id cannot be recycled
        return tm

    def first_order_transfer_map(self, energy: torch.Tensor, species: Species):
        if self.is_skippable:
            return self._compose_run(("all",), list(self.elements), energy, species)
        return None

    # ---- tracking ---------------------------------------------------------------------------------------
    def _apply_run(self, key, run, incoming: ParticleBeam) -> ParticleBeam:
        tm = self._compose_run(key, run, incoming.energy, incoming.species)
        new_particles = _ops.apply_map(incoming.particles, tm)
        length = None
        for e in run:
            length = e.length if length is None else length + e.length
        return ParticleBeam(new_particles, incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=incoming.s + length,
                            species=incoming.species)

    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        elements = list(self.elements)
        if all(e.is_skippable for e in elements):
            return self._apply_run(("all",), elements, incoming)
        run, start = [], 0
        for i, e in enumerate(elements):
            if e.is_skippable:
                if not run:
                    start = i
                run.append(e)
            else:
                if run:
                    incoming = self._apply_run((start, i), run, incoming)
                    run = []
                incoming = e.track(incoming)
        if run:
            incoming = self._apply_run((start, len(elements)), run, incoming)
        return incoming

    def track_elementwise(self, incoming: ParticleBeam, fused: bool = False) -> ParticleBeam:
        """Track element by element WITHOUT merging transfer maps (every element is a real pass over
        the particles, results identical to `for e in elements: beam = e.track(beam)`). Runs of linear
        elements are dispatched as one `chx_track_elementwise` (E passes over HBM) or, with
        `fused=True`, one `chx_track_fused` call (one pass, particle kept in registers)."""
        elements = list(self.elements)
        run: list[Element] = []
        cache = self.__dict__["_run_cache"]

        def flush(beam):
            if not run:
                return beam
            # the stacked [E][B][7][7] map table of a run is cached like the composed map
            key = ("stack", id(run[0]), len(run))
            cacheable = not (beam.energy.requires_grad or beam.species.mass_eV.requires_grad)
            token = self._run_token(run, beam.energy, beam.species) if cacheable else None
            hit = cache.get(key)
            if cacheable and hit is not None and hit[0] == token:
                stack = hit[1]
            else:
                maps = [e.first_order_transfer_map(beam.energy, beam.species) for e in run]
                bshape = torch.broadcast_shapes(beam.energy.shape, *[m.shape[:-2] for m in maps])
                Bm = _ops.numel(bshape)
                stack = torch.stack([m.expand(*bshape, 7, 7).reshape(Bm, 7, 7) for m in maps])
                if cacheable and not stack.requires_grad:
                    cache[key] = (token, stack, beam.energy)
            out = _ops.track_elementwise(beam.particles, stack, fused=fused)
            length = None
            for e in run:
                length = e.length if length is None else length + e.length
            run.clear()
            return ParticleBeam(out, beam.energy, particle_charges=beam.particle_charges,
                                survival_probabilities=beam.survival_probabilities, s=beam.s + length,
                                species=beam.species)

        for e in elements:
            if e.is_skippable and not isinstance(e, Segment):
                run.append(e)
            else:
                incoming = flush(incoming)
                incoming = e.track_elementwise(incoming, fused) if isinstance(e, Segment) else e.track(incoming)
        return flush(incoming)

    def get_beam_attrs_along_segment(self, attr_names, incoming: ParticleBeam, resolution=None):
        """Beam attributes after every element (segment.py:658-700)."""
        single = isinstance(attr_names, str)
        names = (attr_names,) if single else tuple(attr_names)
        beams = [incoming]
        for e in self.elements:
            beams.append(e.track(beams[-1]))
        results = tuple(torch.stack(torch.broadcast_tensors(*[getattr(b, n) for b in beams]), dim=-1) for n in names)
        return results[0] if single else results

    # ---- lattice utilities (segment.py:179-367) -------------------------------------------------------------
    def flattened(self) -> "Segment":
        flat = []
        for e in self.elements:
            flat += list(e.flattened().elements) if isinstance(e, Segment) else [e]
        return Segment(flat, name=self.name)

    def transfer_maps_merged(self, incoming_beam: ParticleBeam, except_for=None) -> "Segment":
        """Merge runs of skippable elements into CustomTransferMaps (segment.py:179-229)."""
        from .custom_transfer_map import CustomTransferMap

        except_for = except_for or []
        merged, run = [], []
        beam = incoming_beam

        def flush():
            nonlocal run, beam
            if len(run) > 1:
                merged.append(CustomTransferMap.from_merging_elements(run, beam))
            elif run:
                merged.append(run[0])
            for e in run:
                beam = e.track(beam)
            run = []

        for e in self.elements:
            if e.is_skippable and e.name not in except_for:
                run.append(e)
            else:
                flush()
                merged.append(e)
                beam = e.track(beam)
        flush()
        return Segment(merged, name=self.name)

    def clone(self) -> "Segment":
        return self.__class__(elements=[e.clone() for e in self.elements], name=self.name,
                              metadata=deepcopy(self.metadata))

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["elements"]

    def __repr__(self) -> str:
        return f"Segment(elements={list(self.elements)!r}, name={self.name!r})"

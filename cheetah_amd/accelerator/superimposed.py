"""Superimposed (mirror of cheetah/accelerator/superimposed.py:14-98): a zero-length element placed at the centre of a
base element = Segment([first half, superimposed element, second half]); everything is delegated to that segment, so
the halves are merged / tracked by the same kernels as any other lattice."""

from __future__ import annotations

import torch

from .element import Element
from .segment import Segment


class Superimposed(Element):
    def __init__(self, base_element: Element, superimposed_element: Element, name=None, sanitize_name=None,
                 metadata=None, device=None, dtype=None):
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)
        del self._buffers["length"]  # derived from the inner segment
        assert bool((superimposed_element.length == 0.0).all()), "The superimposed element must have zero length."
        self.base_element = base_element
        self.superimposed_element = superimposed_element
        halves = base_element.split(base_element.length / 2.0)
        assert len(halves) == 2, f"{type(base_element).__name__} cannot be split into two halves"
        self._segment = Segment(elements=[halves[0], superimposed_element, halves[1]], name=f"{self.name}_segment")

    def flattened(self) -> Segment:
        return self._segment.flattened()

    @property
    def is_skippable(self) -> bool:
        return self._segment.is_skippable

    @property
    def length(self) -> torch.Tensor:
        return self._segment.length

    def first_order_transfer_map(self, energy, species):
        return self._segment.first_order_transfer_map(energy, species)

    def track(self, incoming):
        return self._segment.track(incoming)

    def clone(self) -> "Superimposed":
        return self.__class__(self.base_element.clone(), self.superimposed_element.clone(), name=self.name)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["base_element", "superimposed_element"]

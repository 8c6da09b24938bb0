"""Corrector magnets (mirror of cheetah/accelerator/horizontal_corrector.py:39-78,
vertical_corrector.py:39-78, combined_corrector.py:41-98): drift map + affine kick in column 6."""

from __future__ import annotations

import torch

from .. import _ops
from .element import Element


class HorizontalCorrector(Element):
    supported_tracking_methods = ["linear"]
    _chx_kind = _ops.KIND["hcor"]

    def __init__(self, length, angle=None, name=None, sanitize_name=None, metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        self.register_buffer_or_parameter("angle", angle if angle is not None else torch.tensor(0.0, **fk))

    def _builder_params(self):
        return [self.length, self.angle]

    def _builder_scalar_refs(self):
        return [(t, None) for t in self._settings("length", "angle")]

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def is_active(self) -> bool:
        return bool((self.angle != 0).any().item())

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "angle"]


class VerticalCorrector(HorizontalCorrector):
    _chx_kind = _ops.KIND["vcor"]


class CombinedCorrector(Element):
    supported_tracking_methods = ["linear"]
    _chx_kind = _ops.KIND["ccor"]

    def __init__(self, length, horizontal_angle=None, vertical_angle=None, name=None, sanitize_name=None,
                 metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        z = lambda v: v if v is not None else torch.tensor(0.0, **fk)  # noqa: E731
        self.register_buffer_or_parameter("horizontal_angle", z(horizontal_angle))
        self.register_buffer_or_parameter("vertical_angle", z(vertical_angle))

    def _builder_params(self):
        return [self.length, self.horizontal_angle, self.vertical_angle]

    def _builder_scalar_refs(self):
        return [(t, None) for t in self._settings("length", "horizontal_angle", "vertical_angle")]

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def is_active(self) -> bool:
        return bool(((self.horizontal_angle != 0) | (self.vertical_angle != 0)).any().item())

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "horizontal_angle", "vertical_angle"]

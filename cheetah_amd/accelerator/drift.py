"""Drift (mirror of cheetah/accelerator/drift.py:40-158: linear, second_order and drift_kick_drift tracking)."""

from __future__ import annotations

from .. import _ops
from .element import Element


class Drift(Element):
    """Drift section in a particle accelerator."""

    supported_tracking_methods = ["linear", "second_order", "drift_kick_drift"]
    _chx_kind = _ops.KIND["drift"]
    _dkd_kind = _ops.DKD_KIND["drift"]
    _t_kind = _ops.T_KIND["drift"]

    def __init__(self, length, tracking_method="linear", name=None, sanitize_name=None, metadata=None,
                 device=None, dtype=None) -> None:
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)
        self.length = length
        self.tracking_method = tracking_method

    def _builder_params(self):
        return [self.length]

    def _builder_scalar_refs(self):
        return [(t, None) for t in self._settings("length")]

    @property
    def is_skippable(self) -> bool:
        return self.tracking_method == "linear"

    _merge_equal = ("tracking_method",)

    def split(self, resolution):
        return self._split_evenly(resolution)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length"]

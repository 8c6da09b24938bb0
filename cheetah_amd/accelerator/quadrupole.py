"""Quadrupole (mirror of cheetah/accelerator/quadrupole.py:52-255: linear, second_order, drift_kick_drift)."""

from __future__ import annotations

import torch

from .. import _ops
from .element import Element


class Quadrupole(Element):
    """Quadrupole magnet: R = R_exit @ base_rmatrix(L, k1, 0) @ R_entry (tilt + misalignment)."""

    supported_tracking_methods = ["linear", "second_order", "drift_kick_drift"]
    _chx_kind = _ops.KIND["quadrupole"]
    _dkd_kind = _ops.DKD_KIND["quadrupole"]
    _t_kind = _ops.T_KIND["quadrupole"]

    def __init__(self, length, k1=None, misalignment=None, tilt=None, num_steps=1, tracking_method="linear",
                 name=None, sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        self.register_buffer_or_parameter("k1", k1 if k1 is not None else torch.tensor(0.0, **fk))
        self.register_buffer_or_parameter(
            "misalignment", misalignment if misalignment is not None else torch.tensor((0.0, 0.0), **fk))
        self.register_buffer_or_parameter("tilt", tilt if tilt is not None else torch.tensor(0.0, **fk))
        self.num_steps = num_steps
        self.tracking_method = tracking_method

    def _builder_params(self):
        return [self.length, self.k1, self.tilt, self.misalignment[..., 0], self.misalignment[..., 1]]

    def _builder_scalar_refs(self):
        length, k1, tilt, m = self._settings("length", "k1", "tilt", "misalignment")
        return [(length, None), (k1, None), (tilt, None), (m, 0), (m, 1)]

    def _dkd_options(self):
        return int(self.num_steps), 3

    @property
    def is_skippable(self) -> bool:
        return self.tracking_method == "linear"

    @property
    def is_active(self) -> bool:
        return bool((self.k1 != 0).any().item())

    _merge_equal = ("tracking_method", "misalignment", "tilt")
    _merge_weighted = ("k1",)
    _merge_summed = ("num_steps",)

    def split(self, resolution):
        return self._split_evenly(resolution)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "k1", "misalignment", "tilt", "num_steps"]

"""Remaining linear elements of SURVEY section 8 row f3 (mirror of cheetah/accelerator/solenoid.py:41-116,
undulator.py:41-125, sextupole.py:47-88): their first-order maps come from the same `chx_build_rmatrix`
kernel (kinds SOLENOID / UNDULATOR / DRIFT) and are tracked by the same apply kernel."""

from __future__ import annotations

import torch

from .. import _ops
from .element import Element


class Solenoid(Element):
    """Hard-edge solenoid with transverse misalignment."""

    supported_tracking_methods = ["linear"]
    _chx_kind = _ops.KIND["solenoid"]

    def __init__(self, length, k=None, misalignment=None, name=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        self.register_buffer_or_parameter("k", k if k is not None else torch.tensor(0.0, **fk))
        self.register_buffer_or_parameter(
            "misalignment", misalignment if misalignment is not None else torch.tensor((0.0, 0.0), **fk))

    def _builder_params(self):
        return [self.length, self.k, self.misalignment[..., 0], self.misalignment[..., 1]]

    def _builder_scalar_refs(self):
        m = self.misalignment
        return [(self.length, None), (self.k, None), (m, 0), (m, 1)]

    _merge_equal = ("misalignment",)
    _merge_weighted = ("k",)

    def split(self, resolution):
        return self._split_evenly(resolution)

    @property
    def is_active(self) -> bool:
        return bool((self.k != 0).any().item())

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "k", "misalignment"]


class Undulator(Element):
    """Planar / helical undulator (linear focusing only, no radiation)."""

    supported_tracking_methods = ["linear"]
    _chx_kind = _ops.KIND["undulator"]

    def __init__(self, length, period=None, kx=None, ky=None, name=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        self.register_buffer_or_parameter("kx", kx if kx is not None else torch.tensor(0.0, **fk))
        self.register_buffer_or_parameter("ky", ky if ky is not None else torch.tensor(0.0, **fk))
        self.register_buffer_or_parameter("period", period if period is not None else torch.tensor(1.0, **fk))

    def _builder_params(self):
        return [self.length, self.kx, self.ky, self.period]

    @property
    def is_active(self) -> bool:
        return bool(torch.logical_or(self.kx != 0.0, self.ky != 0.0).any().item())

    @property
    def is_skippable(self) -> bool:
        return True

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "period", "kx", "ky"]


class Sextupole(Element):
    """Sextupole (sextupole.py:45-132): a drift in first order, the k2 kick through the second-order tensor."""

    supported_tracking_methods = ["linear", "second_order"]
    _chx_kind = _ops.KIND["drift"]
    _t_kind = _ops.T_KIND["sextupole"]

    def __init__(self, length, k2=None, misalignment=None, tilt=None, tracking_method="second_order", name=None,
                 sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.length = length
        self.register_buffer_or_parameter("k2", k2 if k2 is not None else torch.tensor(0.0, **fk))
        self.register_buffer_or_parameter(
            "misalignment", misalignment if misalignment is not None else torch.tensor((0.0, 0.0), **fk))
        self.register_buffer_or_parameter("tilt", tilt if tilt is not None else torch.tensor(0.0, **fk))
        self.tracking_method = tracking_method

    def _builder_params(self):
        return [self.length]

    def _t_params(self):
        return [self.length, self.k2, self.tilt, self.misalignment[..., 0], self.misalignment[..., 1]]

    _merge_equal = ("tracking_method", "k2", "misalignment", "tilt")

    @property
    def is_active(self) -> bool:
        return bool((self.k2 != 0).any().item())

    @property
    def is_skippable(self) -> bool:
        return self.tracking_method == "linear"

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "k2", "misalignment", "tilt"]


class TransverseDeflectingCavity(Element):
    """Transverse deflecting cavity (transverse_deflecting_cavity.py:44-209): half drift, transverse RF kick with the
    matching energy change, half drift, tracked per particle by chx_dkd_track (drift_kick_drift only)."""

    supported_tracking_methods = ["drift_kick_drift"]
    _dkd_kind = _ops.DKD_KIND["tdc"]

    def __init__(self, length, voltage=None, phase=None, frequency=None, misalignment=None, tilt=None, num_steps=1,
                 tracking_method="drift_kick_drift", name=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        z = lambda v: v if v is not None else torch.tensor(0.0, **fk)  # noqa: E731
        self.length = length
        self.register_buffer_or_parameter("voltage", z(voltage))
        self.register_buffer_or_parameter("phase", z(phase))
        self.register_buffer_or_parameter("frequency", z(frequency))
        self.register_buffer_or_parameter(
            "misalignment", misalignment if misalignment is not None else torch.tensor((0.0, 0.0), **fk))
        self.register_buffer_or_parameter("tilt", z(tilt))
        self.num_steps = num_steps
        self._tracking_method = "drift_kick_drift"
        self.tracking_method = tracking_method

    def _dkd_params(self):
        return [self.length, self.voltage, self.phase, self.frequency, self.tilt, self.misalignment[..., 0],
                self.misalignment[..., 1]]

    @property
    def is_active(self) -> bool:
        return bool((self.voltage != 0).any().item())

    @property
    def is_skippable(self) -> bool:
        return False

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "voltage", "phase", "frequency", "misalignment", "tilt",
                                            "num_steps"]

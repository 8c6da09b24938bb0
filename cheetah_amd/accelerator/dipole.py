"""Dipole and RBend (mirror of cheetah/accelerator/dipole.py:58-135,372-466 and rbend.py:49-116)."""

from __future__ import annotations

import torch

from .. import _ops
from .element import Element


class Dipole(Element):
    """Sector bend: R = rot^T (R_exit_face @ base_rmatrix(L, k1, angle/L) @ R_enter_face) rot."""

    supported_tracking_methods = ["linear", "second_order", "drift_kick_drift"]
    _chx_kind = _ops.KIND["dipole"]
    _dkd_kind = _ops.DKD_KIND["dipole"]
    _t_kind = _ops.T_KIND["dipole"]

    def __init__(self, length, angle=None, k1=None, dipole_e1=None, dipole_e2=None, tilt=None, gap=None,
                 gap_exit=None, fringe_integral=None, fringe_integral_exit=None, fringe_at="both",
                 fringe_type="linear_edge", tracking_method="linear", name=None, sanitize_name=None,
                 metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        z = lambda v: v if v is not None else torch.tensor(0.0, **fk)  # noqa: E731
        self.length = length
        self.register_buffer_or_parameter("angle", z(angle))
        self.register_buffer_or_parameter("k1", z(k1))
        self.register_buffer_or_parameter("_e1", z(dipole_e1))
        self.register_buffer_or_parameter("_e2", z(dipole_e2))
        self.register_buffer_or_parameter("fringe_integral", z(fringe_integral))
        self.register_buffer_or_parameter(
            "fringe_integral_exit", fringe_integral_exit if fringe_integral_exit is not None else self.fringe_integral)
        self.register_buffer_or_parameter("gap", z(gap))
        self.register_buffer_or_parameter("gap_exit", gap_exit if gap_exit is not None else self.gap)
        self.register_buffer_or_parameter("tilt", z(tilt))
        self.fringe_at = fringe_at
        self.fringe_type = fringe_type
        self.tracking_method = tracking_method

    @property
    def hx(self) -> torch.Tensor:
        return self.angle / self.length

    @property
    def dipole_e1(self) -> torch.Tensor:
        return self._e1

    @dipole_e1.setter
    def dipole_e1(self, value) -> None:
        self._e1 = value
        self._touch()

    @property
    def dipole_e2(self) -> torch.Tensor:
        return self._e2

    @dipole_e2.setter
    def dipole_e2(self, value) -> None:
        self._e2 = value
        self._touch()

    def _builder_params(self):
        # NB: like the reference (dipole.py:453-459) the exit face uses `gap`, not `gap_exit`
        return [self.length, self.angle, self.k1, self._e1, self._e2, self.tilt, self.fringe_integral,
                self.fringe_integral_exit, self.gap]

    def _dkd_params(self):
        # dipole.py:183-370 (Bmad-X body + linear_edge fringes); unlike the linear map the exit face uses gap_exit
        return [self.length, self.angle, self._e1, self._e2, self.tilt, self.fringe_integral,
                self.fringe_integral_exit, self.gap, self.gap_exit]

    def _dkd_scalar_refs(self):
        return [(t, None) for t in self._dkd_params()]       # (buffers of the element, RBend's derived face angles included)

    def _dkd_options(self):
        return 1, _ops.FRINGE_AT[self.fringe_at]

    @property
    def is_skippable(self) -> bool:
        return self.tracking_method == "linear"

    @property
    def is_active(self) -> bool:
        return bool((self.angle != 0).any().item())

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "angle", "k1", "dipole_e1", "dipole_e2", "tilt", "gap",
                                            "gap_exit", "fringe_integral", "fringe_integral_exit", "fringe_at",
                                            "fringe_type"]


class RBend(Dipole):
    """Rectangular bend: a Dipole whose pole-face angles include half the bend angle (rbend.py:104-116)."""

    def __init__(self, length, angle=None, k1=None, rbend_e1=None, rbend_e2=None, tilt=None, gap=None,
                 gap_exit=None, fringe_integral=None, fringe_integral_exit=None, fringe_at="both",
                 fringe_type="linear_edge", tracking_method="linear", name=None, sanitize_name=None,
                 metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        angle = angle if angle is not None else torch.tensor(0.0, **fk)
        e1 = rbend_e1 if rbend_e1 is not None else torch.tensor(0.0, **fk)
        e2 = rbend_e2 if rbend_e2 is not None else torch.tensor(0.0, **fk)
        super().__init__(length=length, angle=angle, k1=k1, dipole_e1=e1 + angle / 2, dipole_e2=e2 + angle / 2,
                         tilt=tilt, gap=gap, gap_exit=gap_exit, fringe_integral=fringe_integral,
                         fringe_integral_exit=fringe_integral_exit, fringe_at=fringe_at, fringe_type=fringe_type,
                         tracking_method=tracking_method, name=name, sanitize_name=sanitize_name,
                         metadata=metadata, **fk)

    @property
    def rbend_e1(self) -> torch.Tensor:
        return self._e1 - self.angle / 2

    @rbend_e1.setter
    def rbend_e1(self, value: torch.Tensor) -> None:  # rbend.py:107-110
        self.dipole_e1 = value + self.angle / 2

    @property
    def rbend_e2(self) -> torch.Tensor:
        return self._e2 - self.angle / 2

    @rbend_e2.setter
    def rbend_e2(self, value: torch.Tensor) -> None:  # rbend.py:114-117
        self.dipole_e2 = value + self.angle / 2

    @property
    def defining_features(self) -> list[str]:
        feats = [f for f in super().defining_features if f not in ("dipole_e1", "dipole_e2")]
        return feats + ["rbend_e1", "rbend_e2"]

"""Pass-through elements (mirror of cheetah/accelerator/marker.py:45-57, bpm.py:66-87,
aperture.py:79-135): identity map, skippable unless active.

`track` keeps the reference's contract — the returned beam is a deep copy (`incoming.clone()`, marker.py:52-53, bpm.py:87):
editing it in place never reaches the incoming beam. `Segment.track` walks its elements through `_track_internal`, which
hands the same tensors on (nothing inside the walk edits a beam in place), and un-aliases once at the end (segment.py)."""

from __future__ import annotations

import torch

from .. import _ops
from ..particles.particle_beam import ParticleBeam
from .._cache import TensorKey
from .element import Element


def _unaliased(outgoing, incoming):
    """`outgoing` if it carries its own coordinate tensor, else a deep copy (the reference's `incoming.clone()`). The
    pass-through elements hand on the very tensor object (`Beam._view`), so identity is the test."""
    a = outgoing.particles if isinstance(outgoing, ParticleBeam) else getattr(outgoing, "mu", None)
    b = incoming.particles if isinstance(incoming, ParticleBeam) else getattr(incoming, "mu", None)
    if outgoing is incoming or (a is not None and a is b):
        return outgoing.clone()
    return outgoing


class Marker(Element):
    _chx_kind = _ops.KIND["identity"]

    def __init__(self, name=None, sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)

    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        return incoming._view()

    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        return _unaliased(self._track_internal(incoming), incoming)

    @property
    def is_skippable(self) -> bool:
        return True


class BPM(Marker):
    """Beam position monitor: reads (mu_x, mu_y) of the passing beam when active (bpm.py:77-87)."""

    _is_bpm = True

    def __init__(self, is_active=False, name=None, misalignment=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.is_active = is_active
        self.register_buffer_or_parameter("misalignment",
                                          misalignment if misalignment is not None else torch.zeros(2, **fk))
        # (x, y) of the last tracked beam relative to the monitor; NaN until a beam has been read (bpm.py:57-61)
        self.register_buffer("reading", torch.tensor((float("nan"), float("nan")), **fk), persistent=False)

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        if self.is_active:
            # both means come out of one fused chx_moments call (bpm.py:77-85). The reading is an OUTPUT of tracking, not a
            # setting: it goes straight into the buffer table — an attribute assignment would move the process-wide epoch and
            # have every persistent run plan of the lattice re-validate on its next use (25 active BPMs in a 100-element lattice:
            # 92 us per BPM and track)
            p = getattr(incoming, "particles", None)
            from .. import sharding

            group = sharding.active_group() if p is not None else None
            if group is not None:
                # the beam's particles are spread over the ranks of a process group (sharding.particle_sharded): the monitor reads
                # the mean of ALL of them — this rank's one-pass moments, one all-gather of 29 doubles, the exact merge
                # (cached per version of the beam: shared with its properties). The stored reading carries no graph — the
                # exchange runs under no_grad whatever the beam carries (a differentiable track through an active BPM
                # must not need one)
                with torch.no_grad():
                    xy = incoming._global_moments(group)[..., 2:5:2].to(p.dtype)
            elif p is not None and not (torch.is_grad_enabled() and (p.requires_grad or incoming.survival_probabilities.requires_grad)):
                # entries 2 and 4 of the moment vector = (mu_x, mu_y): one strided view, one cast, one subtraction
                xy = incoming._moments()[..., 2:5:2].to(p.dtype)
            else:
                xy = torch.stack([incoming.mu_x, incoming.mu_y], dim=-1)
            self.__dict__["_buffers"]["reading"] = xy - self.misalignment
        return incoming._view()

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["is_active"]


class Aperture(Marker):
    """Aperture: when active, zeroes the survival probability of particles outside
    (aperture.py:90-135). The mask is a per-particle elementwise torch op on the device."""

    _is_aperture = True

    def __init__(self, x_max=None, y_max=None, shape="rectangular", is_active=True, name=None, sanitize_name=None,
                 metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        inf = lambda v: v if v is not None else torch.tensor(float("inf"), **fk)  # noqa: E731
        self.register_buffer_or_parameter("x_max", inf(x_max))
        self.register_buffer_or_parameter("y_max", inf(y_max))
        self.shape = shape
        self.is_active = is_active

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        if self.is_active:   # aperture.py:129-135: a new beam that shares the particle tensor, new survival probabilities
            return self._track_internal(incoming)
        return _unaliased(self._track_internal(incoming), incoming)

    def _check_limits(self) -> None:
        limits = (self.x_max, self.y_max)
        checked = self.__dict__.get("_limits_checked")
        if checked is None or not checked.matches(limits):
            assert bool((self.x_max >= 0).all()) and bool((self.y_max >= 0).all())
            self.__dict__["_limits_checked"] = TensorKey(limits)
        assert self.shape in ["rectangular", "elliptical"], f"Unknown aperture shape {self.shape}"

    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        if not self.is_active:
            return incoming._view()
        if not isinstance(incoming, ParticleBeam):  # aperture.py:94-100
            import warnings

            from .element import PhysicsWarning

            warnings.warn("Aperture tracking is currently only supported for `ParticleBeam`.", PhysicsWarning,
                          stacklevel=2)
            return incoming
        # aperture.py:72-73 asserts non-negative half-widths on every track; here the (host-synchronising) check runs once
        # per value of the two tensors
        self._check_limits()
        # one streaming kernel (chx_aperture_mask): strict `<` for the rectangle, `<= 1` for the ellipse
        survival = _ops.aperture_mask(incoming.particles, incoming.survival_probabilities, self.x_max, self.y_max,
                                      self.shape)
        return ParticleBeam(incoming.particles, incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=survival, s=incoming.s, species=incoming.species)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["x_max", "y_max", "shape", "is_active"]

"""Cavity (mirror of cheetah/accelerator/cavity.py:51-358).

The first-order map comes from the `chx_build_rmatrix` kernel (kinds CAVITY_SW / CAVITY_TW), the
per-batch track coefficients from `chx_cavity_coeffs` (including the reference's data-dependent
`if (delta_energy > 0).any()` branch, evaluated on device) and the particle update is ONE fused
kernel, `chx_cavity_track`: matrix apply + delta / tau rewrite (the reference makes ~25 elementwise
passes over the particle array, cavity.py:112-226).
"""

from __future__ import annotations

import ctypes

import torch

from .. import _lib, _ops
from ..particles.parameter_beam import ParameterBeam
from ..particles.particle_beam import ParticleBeam
from .element import Element, tracking_call


def _narrow_to(t: torch.Tensor, shape) -> torch.Tensor:
    """Pick the representative entries of a broadcast tensor `t` so that it has `shape` (right-aligned, every dim of
    `shape` equals the matching dim of `t` or is 1): the values do not depend on the dropped dims."""
    lead = t.dim() - len(shape)
    if lead:
        t = t[(0,) * lead]
    for d, s in enumerate(shape):
        if s == 1 and t.shape[d] != 1:
            t = t.narrow(d, 0, 1)
    return t.reshape(shape)


class Cavity(Element):
    """Accelerating RF cavity (standing or traveling wave)."""

    _static_skippable = False  # `voltage != 0` is a tensor-value dependent flag
    _is_cavity = True

    def __init__(self, length, voltage=None, phase=None, frequency=None, cavity_type="standing_wave", name=None,
                 sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        z = lambda v: v if v is not None else torch.tensor(0.0, **fk)  # noqa: E731
        self.length = length
        self.register_buffer_or_parameter("voltage", z(voltage))
        self.register_buffer_or_parameter("phase", z(phase))
        self.register_buffer_or_parameter("frequency", z(frequency))
        self.cavity_type = cavity_type   # not validated here: like the reference, an unknown type fails when its map is needed

    @property
    def _chx_kind(self) -> int:
        return _ops.KIND[self._kind_name()]

    def _kind_name(self) -> str:
        if self.cavity_type == "standing_wave":
            return "cavity_sw"
        if self.cavity_type == "traveling_wave":
            return "cavity_tw"
        if not self.is_active:
            return "cavity_sw"       # switched off: the drift-like map, whatever the type says (cavity.py:253-262)
        raise ValueError(f"Invalid cavity type: {self.cavity_type}")   # cavity.py:337

    def _builder_params(self):
        return [self.length, self.voltage, self.phase, self.frequency]

    def _plannable(self) -> bool:
        # a switched-off cavity inside a run (Segment._plan re-examines `is_active` on every track and re-partitions the lattice
        # when the voltage comes back): its drift-like map is built on the device from the four settings like any magnet's —
        # a linac with two of sixteen cavities off was 241 us per track instead of 49
        return not self.is_active

    def _energy_graph(self, e_out: torch.Tensor, energy: torch.Tensor) -> torch.Tensor:
        """The outgoing energy is `energy + voltage * cos(phase) * q` (cavity.py:113-122): it carries a graph only through
        those three. The coefficient expressions are evaluated on the stacked settings, where a trainable length or frequency
        would otherwise mark it as well."""
        if e_out.requires_grad and not (energy.requires_grad or self.voltage.requires_grad or self.phase.requires_grad):
            return e_out.detach()
        return e_out

    def _energy_shape(self, energy: torch.Tensor):
        """Shape of the outgoing energy: `incoming.energy + voltage * cos(phase) * ...` (cavity.py:113-122)."""
        return torch.broadcast_shapes(self.voltage.shape, self.phase.shape, energy.shape)

    @property
    def is_active(self) -> bool:
        # host-side flag, refreshed only when the voltage tensor changes (one sync per change)
        # (the tensor itself is kept in the entry: an `id` could be recycled by a later tensor)
        v = self.voltage
        cached = self.__dict__.get("_active_cache")
        if cached is None or cached[0] is not v or cached[1] != v._version:
            cached = (v, v._version, bool((v != 0).any().item()))
            self.__dict__["_active_cache"] = cached
        return cached[2]

    @property
    def is_skippable(self) -> bool:
        return not self.is_active

    def _track_parameter_beam(self, incoming):
        """cavity.py:108-110,127-133,202-218,229-239: matrix part plus the reference's moment updates, one kernel."""
        dtype = incoming.mu.dtype
        tm = self.first_order_transfer_map(incoming.energy, incoming.species)
        tensors = [t.to(dtype) for t in self._builder_params()]
        energy = incoming.energy.to(dtype)
        _ops.require_device(incoming.mu, energy, *tensors)
        pshape = torch.broadcast_shapes(*[t.shape for t in tensors])
        batch_shape = torch.broadcast_shapes(pshape, energy.shape, incoming.mu.shape[:-1], incoming.cov.shape[:-2])
        B = _ops.numel(batch_shape)
        if len(pshape) == 0:
            params = torch.stack(tensors).reshape(1, 4)
        else:
            params = torch.stack([t.expand(batch_shape) for t in tensors], dim=-1).reshape(B, 4)
        e = energy.reshape(1) if energy.dim() == 0 else energy.expand(batch_shape).reshape(B).contiguous()
        sp = incoming.species
        coeffs, e_out = _ops.cavity_coeffs(params.contiguous(), e, sp.mass_eV_float, sp.num_elementary_charges_float, B)
        e_out = _narrow_to(self._energy_graph(e_out, energy).reshape(batch_shape), self._energy_shape(energy))
        return incoming._tracked(tm, self.length, cavity_coeffs=coeffs, energy=e_out, batch_shape=batch_shape)

    @tracking_call
    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        ref = incoming.mu if isinstance(incoming, ParameterBeam) else getattr(incoming, "particles", None)
        if ref is not None and ref.dtype != self.length.dtype:
            # cavity.py:108-111 multiplies the beam by the element's map: torch refuses mixed float32 / float64 operands
            raise RuntimeError(f"expected m1 and m2 to have the same dtype, but got: {ref.dtype} != {self.length.dtype} "
                               "(beam vs cavity settings)")
        if isinstance(incoming, ParameterBeam):
            return self._track_parameter_beam(incoming)
        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        fast = self._track_scalars(incoming)
        if fast is not None:
            return fast
        dtype = incoming.particles.dtype
        tm = self.first_order_transfer_map(incoming.energy, incoming.species)
        tensors = [t.to(dtype) for t in self._builder_params()]
        energy = incoming.energy.to(dtype)
        _ops.require_device(incoming.particles, energy, *tensors)
        pshape = torch.broadcast_shapes(*[t.shape for t in tensors])
        vshape = torch.broadcast_shapes(pshape, energy.shape)  # shape of the outgoing energy (cavity.py:122)
        batch_shape = torch.broadcast_shapes(vshape, incoming.particles.shape[:-2])
        B, N = _ops.numel(batch_shape), incoming.particles.shape[-2]
        if len(pshape) == 0:
            params = torch.stack(tensors).reshape(1, 4)
        else:
            params = torch.stack([t.expand(batch_shape) for t in tensors], dim=-1).reshape(B, 4)
        e = energy.reshape(1) if energy.dim() == 0 else energy.expand(batch_shape).reshape(B).contiguous()
        sp = incoming.species
        coeffs, e_out = _ops.cavity_coeffs(params.contiguous(), e, sp.mass_eV_float,
                                           sp.num_elementary_charges_float, B)
        x, _ = _ops.flat_bcast(incoming.particles, batch_shape, 2)
        R, _ = _ops.flat_bcast(tm, batch_shape, 2)
        R = R.expand(B, 7, 7).contiguous()
        out = _ops.cavity_track(_ops.aligned(x), R, coeffs, B, N)
        # outgoing energy has the broadcast shape of (voltage, phase, energy) — not of the length or the particles
        e_out = _narrow_to(self._energy_graph(e_out, energy).reshape(batch_shape), self._energy_shape(energy))
        return ParticleBeam(out.reshape(*batch_shape, N, 7), e_out, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=incoming.s + self.length,
                            species=incoming.species)

    def _track_scalars(self, incoming: ParticleBeam):
        """One beam through a cavity whose settings are device scalars of the beam's dtype: the whole element — map,
        coefficients, outgoing energy, particle pass — in ONE C call (chx_cavity_track_scalars, two launches), bit-identical to
        the general path below. None when that does not apply (vectorised beam or settings, mixed dtypes, gradients)."""
        p, e = incoming.particles, incoming.energy
        if p.dim() != 2 or not p.is_cuda or e.dim() != 0 or e.dtype != p.dtype or e.device != p.device:
            return None
        settings = self._settings("length", "voltage", "phase", "frequency")
        for t in settings:
            if t.dim() != 0 or t.dtype != p.dtype or t.device != p.device:
                return None
        sp = incoming.species
        if torch.is_grad_enabled() and (p.requires_grad or e.requires_grad or sp.mass_eV.requires_grad
                                        or sp.num_elementary_charges.requires_grad or any(t.requires_grad for t in settings)):
            return None
        lib = _lib.lib()
        x = p if p.is_contiguous() and p.data_ptr() % 16 == 0 else _ops.aligned(p)
        out = torch.empty_like(x)
        e_out = torch.empty((), dtype=p.dtype, device=p.device)
        # map and coefficients live in a workspace that belongs to this element (consumed by the launch that follows on the
        # same stream; elements, like the reference's caches, are not re-entrant)
        ws = self.__dict__.get("_scalar_ws")
        if ws is None or ws.device != p.device:
            ws = self.__dict__["_scalar_ws"] = torch.empty(lib.chx_cavity_track_scalars_workspace_bytes() // 8 + 1,
                                                           dtype=torch.float64, device=p.device)
        _ops.check_current_device(p.device)
        kind = _ops.KIND[self._kind_name()]
        s_in = incoming.s
        s_out = torch.empty_like(s_in) if (s_in.dim() == 0 and s_in.dtype == p.dtype and s_in.device == p.device
                                           and not s_in.requires_grad) else None
        _ops.check(lib.chx_cavity_track_scalars(x.data_ptr(), (ctypes.c_void_p * 4)(*[t.data_ptr() for t in settings]), e.data_ptr(),
                                                kind, sp.mass_eV_float, sp.num_elementary_charges_float, x.shape[0],
                                                _ops.dtype_code(p.dtype), out.data_ptr(), e_out.data_ptr(),
                                                s_in.data_ptr() if s_out is not None else None,
                                                s_out.data_ptr() if s_out is not None else None, ws.data_ptr(), ws.numel() * 8,
                                                _ops.stream_ptr()), "chx_cavity_track_scalars")
        return ParticleBeam(out, e_out, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities,
                            s=s_out if s_out is not None else s_in + self.length, species=sp)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["length", "voltage", "phase", "frequency", "cavity_type"]

"""Element base class (mirror of cheetah/accelerator/element.py:17-490 for the linear hot path).

`first_order_transfer_map` is produced by the `chx_build_rmatrix` HIP kernel from the element's
parameter tensors (no host round trip) and memoised with the reference's cache rules
(cheetah/utils/cache.py:6-68): key = (id, _version, requires_grad) of every defining tensor plus
the non-tensor features; bypass when energy / species require grad. `track` applies the map with
the `chx_apply_affine7` kernel (element.py:180-191).
"""

from __future__ import annotations

import threading
import warnings
from copy import deepcopy
from typing import Any

import torch
from torch import nn

from .. import _ops
from .._cache import TensorKey
from ..particles.parameter_beam import ParameterBeam
from ..particles.particle_beam import ParticleBeam
from ..particles.species import Species


from ..warnings import DirtyNameWarning, PhysicsWarning  # noqa: E402,F401


_name_counter = 0


def merge_element_names(*names: str, use_shared_prefix: bool = True) -> str:
    """Name of a merged element (utils/names.py:17-38): the shared prefix if there is one, else the names joined."""
    import os

    assert len(names) > 0, "At least one name must be provided."
    common_prefix = os.path.commonprefix(list(names))
    return common_prefix if use_shared_prefix and len(common_prefix) > 0 else "_".join(names)


def _unique_name() -> str:
    global _name_counter
    _name_counter += 1
    return f"unnamed_element_{_name_counter}"


class _TrackDepth(threading.local):
    """> 0 while a tracking call is running ON THIS THREAD: map caches then never compare energies by value (see
    `_cached_map`). Thread-local: a track in one thread must not change the cache semantics of a direct
    `first_order_transfer_map` call in another."""

    value = 0


_TRACK_DEPTH = _TrackDepth()


def tracking_call(fn):
    """Marks a method as a tracking entry point (`_TRACK_DEPTH`)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        depth = _TRACK_DEPTH
        depth.value += 1
        try:
            return fn(*args, **kwargs)
        finally:
            depth.value -= 1

    return wrapper


class Element(nn.Module):
    """Base class of all beamline elements."""

    #: declared by the elements that offer a choice; all others get `[<class name>.lower()]` in __init__, like the
    #: reference (element.py:63-65) — e.g. a Marker does not "support" the linear method, it has none to choose from
    #: libchx map-builder kind (include/chx.h `chx_kind`); None for elements with explicit maps
    _chx_kind: int | None = None
    _is_cavity = False            # Cavity: may join a stretch of `chx_lattice_track` (Segment._lattice_stretch)
    _is_bpm = False               # BPM (when active): likewise, as an item that reads the beam and lets it pass
    _is_aperture = False          # Aperture (when active): likewise, as an item that thins the survival probabilities
    _is_screen = False            # Screen (when active): likewise, as an item that records the beam (and its image) and lets it pass
    #: True when `is_skippable` depends on non-tensor attributes only (those bump `_revision` when set)
    _static_skippable = True
    #: process-wide count of attribute assignments on any element (see `_touch`)
    _epoch = 0
    #: the epoch of the last assignment that may have changed more than an ADDRESS: anything but a plain tensor put in the place
    #: of a registered buffer of the same dtype, device and shape, neither of them carrying a graph (`__setattr__`). What depends
    #: on the lattice's structure only (which elements are skippable, `Segment._plan`) stands while this one stands still; plans
    #: that hold addresses are patched on the spot where they registered for it (`_hooks`, segment._FastRun.absorb) — the control
    #: loop that assigns its actions as new tensors every step (README.md:73-77 of the reference) re-derives nothing.
    _hard_epoch = 0

    def __init__(self, name=None, sanitize_name=None, metadata=None, device=None, dtype=None) -> None:
        super().__init__()
        self.__dict__["_revision"] = 0
        self.name = name if name is not None else _unique_name()
        if not self.name.isidentifier():  # element.py:44-57
            if sanitize_name:
                self.sanitize_name()
            elif sanitize_name is None:
                warnings.warn(
                    f"Dirty element name {self.name} is not a valid Python variable name. You will not be able to use "
                    "the `segment.element_name` syntax to access this element. Set `sanitize_name=True` to change the "
                    "name to a valid one, or `sanitize_name=False` to silence this warning.",
                    category=DirtyNameWarning, stacklevel=2)
        self.metadata = metadata if metadata is not None else {}
        self.register_buffer("length", torch.zeros((), device=device, dtype=dtype))
        if not hasattr(self, "supported_tracking_methods"):
            self.supported_tracking_methods = [self.__class__.__name__.lower()]
        self._tracking_method = self.supported_tracking_methods[0]

    def sanitize_name(self) -> None:
        """Make the name a valid Python identifier: other characters -> '_', leading digit -> '_' prefix."""
        clean = "".join(c if c.isalnum() or c == "_" else "_" for c in self.name)
        self.name = "_" + clean if clean[0].isdigit() else clean

    # ---- parameters packed for the builder kernel, in include/chx.h order ------------------------
    def _builder_params(self) -> list[torch.Tensor]:
        return []

    def _settings(self, *names):
        """The named settings, straight from the buffer dictionary when none of them is a Parameter (nn.Module's
        `__getattr__` costs ~0.4 us per name, and the run plan reads every setting of every touched element per control step)."""
        b = self.__dict__["_buffers"]
        try:
            return [b[n] for n in names]
        except KeyError:
            return [getattr(self, n) for n in names]

    def _plannable(self) -> bool:
        """May this element, standing in a run of skippable elements, go into the run's persistent device plan? Elements whose
        skippability depends on tensor VALUES say no by default."""
        return self._static_skippable

    def _builder_scalar_refs(self):
        """The builder parameters as (tensor, index) pairs for the all-scalar fast path (`_ops.build_compose_scalars`):
        index None = the 0-d tensor itself, an integer = that entry of a 1-d tensor (no view object is created)."""
        return [(t, None) for t in self._builder_params()]

    def _work_dtype(self) -> torch.dtype:
        return self.length.dtype

    def _build_map(self, energy: torch.Tensor, species: Species) -> torch.Tensor:
        """Evaluate the element's 7x7 map on device for `energy` (…) -> (…, 7, 7)."""
        kind = self._chx_kind
        dtype = self._work_dtype()
        tensors = [t.to(dtype) if t.dtype != dtype else t for t in self._builder_params()]
        energy = energy.to(dtype) if energy.dtype != dtype else energy
        _ops.require_device(energy, *tensors)
        pshape = torch.broadcast_shapes(*[t.shape for t in tensors]) if tensors else ()
        shape = torch.broadcast_shapes(pshape, energy.shape)
        B = _ops.numel(shape)
        if not tensors:
            params = None
        elif len(pshape) == 0:
            params = torch.stack(tensors).reshape(1, -1)
        else:
            params = torch.stack([t.expand(shape) for t in tensors], dim=-1).reshape(B, len(tensors))
        e = energy.reshape(1) if energy.dim() == 0 else energy.expand(shape).reshape(B)
        R = _ops.build_rmatrix(kind, params, e, species.mass_eV_float, species.num_elementary_charges_float, B)
        return R.reshape(*shape, 7, 7)

    # ---- cache (utils/cache.py) -----------------------------------------------------------------
    def _feature_key(self):
        """(non-tensor features, defining tensors). The tensors are compared by identity / version / requires_grad against
        a `TensorKey` that keeps them referenced (an `id` alone can be recycled by a later tensor)."""
        static, tensors = [], []
        for name in self.defining_features:
            f = getattr(self, name)
            if isinstance(f, torch.Tensor):
                static.append(None)
                tensors.append(f)
            else:
                static.append(f if not isinstance(f, (list, dict)) else repr(f))
        return tuple(static), tensors

    def first_order_transfer_map(self, energy: torch.Tensor, species: Species) -> torch.Tensor:
        return self._cached_map("_map_cache", self._build_map, energy, species)

    def transfer_map(self, energy: torch.Tensor, species: Species) -> torch.Tensor:
        """Deprecated name of `first_order_transfer_map` (element.py:67-104)."""
        warnings.warn("The `transfer_map` method is deprecated and will be removed in a future version. Use "
                      "`first_order_transfer_map` instead.", DeprecationWarning, stacklevel=2)
        return self.first_order_transfer_map(energy, species)

    def second_order_transfer_map(self, energy: torch.Tensor, species: Species) -> torch.Tensor:
        """T_ijk with x_out_i = sum_jk T_ijk x_j x_k (element.py:134-148), built on device by chx_build_ttensor."""
        if self._t_kind is None:
            raise NotImplementedError
        return self._cached_map("_tmap_cache", self._build_ttensor, energy, species)

    def _t_params(self) -> list[torch.Tensor]:
        return self._builder_params()

    def _build_ttensor(self, energy: torch.Tensor, species: Species) -> torch.Tensor:
        dtype = self._work_dtype()
        energy = energy.to(dtype) if energy.dtype != dtype else energy
        params, pshape = _ops.stack_params(self._t_params(), dtype, energy.device)
        return _ops.build_ttensor(self._t_kind, params, pshape, energy, species.mass_eV_float)

    def _cached_map(self, slot: str, build, energy: torch.Tensor, species: Species) -> torch.Tensor:
        if energy.requires_grad or species.mass_eV.requires_grad or species.num_elementary_charges.requires_grad \
                or _ops.CAPTURING[0]:
            return build(energy, species)
        cache = self.__dict__.get(slot)
        fkey, ftensors = self._feature_key()
        if cache is not None and cache["fkey"] == fkey and cache["tkey"].matches(ftensors) \
                and cache["mass"] == species.mass_eV_float and cache["nq"] == species.num_elementary_charges_float:
            ce = cache["energy_ref"]
            if ce is energy and cache["energy_version"] == energy._version:
                return cache["result"]
            # Another tensor: the reference compares the VALUE (utils/cache.py:47-52, `torch.equal`), which reads the device
            # back — 24 us and a pipeline stall, measured in a linac where every cavity hands on a new energy tensor. A direct
            # call keeps that behaviour (the same map OBJECT for an equal energy); inside a tracking call the map is simply
            # rebuilt, one small launch and no host synchronisation.
            if _TRACK_DEPTH.value == 0 and ce.dtype == energy.dtype and ce.device == energy.device and ce.shape == energy.shape \
                    and torch.equal(cache["energy_copy"], energy):
                return cache["result"]
        result = build(energy, species)
        if result.requires_grad:  # a graph-attached map must not outlive its backward pass
            return result
        self.__dict__[slot] = {
            "fkey": fkey, "tkey": TensorKey(ftensors), "mass": species.mass_eV_float, "nq": species.num_elementary_charges_float,
            "energy_ref": energy, "energy_version": energy._version, "energy_copy": energy.detach().clone(),
            "result": result,
        }
        return result

    # ---- tracking ---------------------------------------------------------------------------------
    @tracking_call
    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        method = self._tracking_method
        if method == "linear":
            return self._track_first_order(incoming)
        if method == "second_order":
            return self._track_second_order(incoming)
        if method == "drift_kick_drift":
            return self._track_drift_kick_drift(incoming)
        raise ValueError(f"Invalid tracking method {method}. For element of type {self.__class__.__name__}, supported "
                         f"methods are {self.supported_tracking_methods}.")

    def _track_second_order(self, incoming: ParticleBeam) -> ParticleBeam:
        """element.py:195-228: one pass of chx_apply_second_order with the element's T tensor."""
        assert isinstance(incoming, ParticleBeam), "Second-order tracking is currently only supported for `ParticleBeam`."
        T = self.second_order_transfer_map(incoming.energy, incoming.species)
        return ParticleBeam(
            _ops.apply_second_order(incoming.particles, T),
            incoming.energy,
            particle_charges=incoming.particle_charges,
            survival_probabilities=incoming.survival_probabilities,
            s=incoming.s + self.length,
            species=incoming.species,
        )

    #: arithmetic of the drift-kick-drift (Bmad-X) kernels for float32 beams (float64 beams are evaluated in float64):
    #:  "mixed" (default)  the longitudinal pair — the (tau, delta) <-> (z, pz) conversions, the z accumulator, the low-energy
    #:                     correction — in float64, everything else in float32 (Drift, Quadrupole on its axis; a
    #:                     misaligned Quadrupole, Dipole and TransverseDeflectingCavity are evaluated in float64). tau and delta as accurate as
    #:                     "double", the transverse coordinates to one float32 rounding per element;
    #:  "double"           every particle in float64, rounded once per element — 1.5 x the time of "mixed" (fp64-VALU bound);
    #:  "storage"          everything in float32 like the reference's own tensor code (cheetah/utils/bmadx.py runs in the beam
    #:                     dtype) — 0.9 x the time, tau / delta lose three to four digits over a lattice.
    #: Measured errors: DESIGN.md section 6. Set on an element (`quad.dkd_precision = "double"`) or on the class for a lattice.
    dkd_precision = "mixed"
    #: chx_dkd_kind / chx_t_kind of the element (include/chx.h); None = method not available
    _dkd_kind: int | None = None
    _t_kind: int | None = None

    def _dkd_params(self) -> list[torch.Tensor]:
        return self._builder_params()

    def _dkd_options(self) -> tuple[int, int]:
        """(num_steps, fringe_at bits) for chx_dkd_track."""
        return 1, 3

    def _dkd_scalar_refs(self):
        """The drift-kick-drift parameters as (tensor, index) pairs like `_builder_scalar_refs`, for elements whose parameters
        are persistent tensors of the element (their version counters tell an edit); None for an element with its own
        `_dkd_params` list that does not say so."""
        if type(self)._dkd_params is not Element._dkd_params:
            return None
        return self._builder_scalar_refs()

    def _dkd_params_stacked(self, dtype, device):
        """The (1, P) parameter array of chx_dkd_track for an element whose settings are all device scalars of `dtype` that
        carry no gradient — kept between tracks (one `torch.stack` = one launch + 6 us of host time per element and track
        otherwise). The key holds the element's revision (any assignment) and the settings' version counters (in-place edits).
        None when the element does not qualify."""
        refs = self._dkd_scalar_refs()
        if refs is None:
            return None
        key = [self.__dict__["_revision"], dtype, device]
        grad = torch.is_grad_enabled()
        for t, index in refs:
            if t.dim() != (0 if index is None else 1) or t.dtype != dtype or t.device != device or (grad and t.requires_grad):
                return None
            key.append(t._version)
        cached = self.__dict__.get("_dkd_cache")
        if cached is not None and cached[0] == key and not _ops.CAPTURING[0]:
            return cached[1]
        with torch.no_grad():
            stacked = torch.stack([t if index is None else t[index] for t, index in refs]).reshape(1, len(refs))
        self.__dict__["_dkd_cache"] = (key, stacked)
        return stacked

    def _track_drift_kick_drift(self, incoming: ParticleBeam) -> ParticleBeam:
        """Bmad-X tracking (e.g. drift.py:106-154): Cheetah -> Bmad coordinates, the element's map and back in one
        kernel pass (chx_dkd_track); the outgoing energy is the reference energy recomputed from p0c."""
        assert isinstance(incoming, ParticleBeam), \
            "Drift-kick-drift tracking is currently only supported for `ParticleBeam`."
        if self.dkd_precision not in _ops.DKD_PRECISION:
            raise ValueError(f"dkd_precision must be 'double', 'mixed' or 'storage', got {self.dkd_precision!r}")
        x, energy = incoming.particles, incoming.energy
        if x.dim() == 2 and x.is_cuda and energy.dim() == 0 and energy.dtype == x.dtype and energy.device == x.device and not (
                torch.is_grad_enabled() and (x.requires_grad or energy.requires_grad)):
            # one plain beam, scalar settings, no graph: the kernel call without the broadcasting / autograd preparations
            params = self._dkd_params_stacked(x.dtype, x.device)
            if params is not None:
                num_steps, fringe = self._dkd_options()
                species = incoming.species
                N = x.shape[0]
                out, e_out = _ops._dkd_raw(self._dkd_kind, _ops.aligned(x).reshape(1, N, 7), params, energy.reshape(1),
                                           species.mass_eV_float, species.num_elementary_charges_float, num_steps, fringe, 1, N,
                                           _ops.DKD_PRECISION[self.dkd_precision])
                return ParticleBeam(out.reshape(N, 7), e_out.reshape(()), particle_charges=incoming.particle_charges,
                                    survival_probabilities=incoming.survival_probabilities, s=incoming.s + self.length,
                                    species=species)
        # the reference's Bmad-X expressions mix the particles with the element's settings: the result is promoted to the wider
        # of the two dtypes (a float64 element gives float64 particles from a float32 beam; the energy keeps the beam's dtype)
        dtype = torch.promote_types(incoming.particles.dtype, self.length.dtype)
        params, pshape = _ops.stack_params(self._dkd_params(), dtype, incoming.particles.device)
        energy = incoming.energy.to(dtype) if incoming.energy.dtype != dtype else incoming.energy
        num_steps, fringe = self._dkd_options()
        species = incoming.species
        x = incoming.particles if incoming.particles.dtype == dtype else incoming.particles.to(dtype)
        particles, ref_energy = _ops.dkd_track(self._dkd_kind, x, params, pshape, energy,
                                               species.mass_eV_float, species.num_elementary_charges_float, num_steps,
                                               fringe, storage_precision=_ops.DKD_PRECISION[self.dkd_precision])
        if ref_energy.dtype != incoming.energy.dtype:
            ref_energy = ref_energy.to(incoming.energy.dtype)
        if ref_energy.requires_grad and not incoming.energy.requires_grad:
            ref_energy = ref_energy.detach()     # recomputed from the incoming reference momentum alone (e.g. drift.py:141-152)
        return ParticleBeam(
            particles,
            ref_energy,
            particle_charges=incoming.particle_charges,
            survival_probabilities=incoming.survival_probabilities,
            s=incoming.s + self.length,
            species=incoming.species,
        )

    def _track_first_order(self, incoming: ParticleBeam) -> ParticleBeam:
        if isinstance(incoming, ParameterBeam):  # element.py:167-179
            tm = self.first_order_transfer_map(incoming.energy, incoming.species)
            return incoming._tracked(tm, self.length)
        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        tm = self.first_order_transfer_map(incoming.energy, incoming.species)
        new_particles = _ops.apply_map(incoming.particles, tm)
        return ParticleBeam(
            new_particles,
            incoming.energy,
            particle_charges=incoming.particle_charges,
            survival_probabilities=incoming.survival_probabilities,
            s=incoming.s + self.length,
            species=incoming.species,
        )

    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        """`track` as called from inside `Segment.track`: pass-through elements override it to hand the incoming tensors on
        without the deep copy their public `track` makes (the segment un-aliases once, at its end)."""
        return self.track(incoming)

    def forward(self, incoming: ParticleBeam) -> ParticleBeam:
        return self.track(incoming)

    # ---- misc API -----------------------------------------------------------------------------------
    @property
    def tracking_method(self) -> str:
        return self._tracking_method

    @tracking_method.setter
    def tracking_method(self, value: str) -> None:
        if value in self.supported_tracking_methods:
            self._tracking_method = value
        else:
            warnings.warn(
                f"Invalid tracking method '{value}' for element {self.name} of type {self.__class__.__name__}, "
                f"supported methods are {self.supported_tracking_methods}. Keeping the previous tracking method "
                f"{self._tracking_method}.", PhysicsWarning, stacklevel=2)

    @property
    def is_skippable(self) -> bool:
        raise NotImplementedError

    def __setattr__(self, name: str, value: Any) -> None:
        d = self.__dict__
        if "_revision" in d and name[0] != "_":
            # a plain tensor assigned to a registered buffer (a control loop's `quad.k1 = new_value`): nn.Module.__setattr__
            # ends up doing exactly this after ~4 us of isinstance checks on Parameter / Module
            if type(value) is torch.Tensor:
                buffers = d.get("_buffers")
                if buffers is not None and name in buffers:
                    old = buffers[name]
                    buffers[name] = value
                    if type(old) is torch.Tensor and old.dtype == value.dtype and old.shape == value.shape and old.device == value.device \
                            and not value.requires_grad and not old.requires_grad and self._static_skippable:
                        # only an address changed: the lattice's structure stands (`_hard_epoch` stays), and the plans that
                        # registered for this element take the new address now instead of re-reading the element at the next track
                        before = Element._epoch
                        self._touch(soft=True)
                        hooks = d.get("_hooks")
                        if hooks:
                            self._absorb(hooks, value, before)
                    else:
                        self._touch()
                    return
            self._touch()
        return super().__setattr__(name, value)

    def _absorb(self, hooks: list, value: torch.Tensor, before: int) -> None:
        """Hand the freshly assigned setting tensor to the plans that hold this element's addresses (`_hooks`: weak references
        left by segment._FastRun.refresh). A plan that was valid at epoch `before` and takes the patch is valid at the new epoch."""
        refs = self._builder_scalar_refs()
        slots = [k for k, (t, index) in enumerate(refs) if t is value]
        rev = self.__dict__["_revision"]
        alive = False
        for ref, i in hooks:
            plan = ref()
            if plan is None:
                continue
            alive = True
            plan.absorb(self, i, refs, slots, rev, before)
        if not alive:
            self.__dict__["_hooks"] = None

    #: derived caches kept in `__dict__`: run plans with raw device addresses (ctypes arrays: not picklable, and a copy would
    #: address the ORIGINAL's tensors), memoised maps / geometry, scratch buffers. A copy or an unpickled element starts
    #: without them and rebuilds on first use.
    _DERIVED_STATE = ("_plan_cache", "_flat_elements", "_map_cache", "_tmap_cache", "_scalar_ws", "_ext_cache",
                      "_grid_tensor", "_geom_cache", "_limits_checked", "_chain_guard_state", "_dkd_cache", "_lattice_cache",
                      "_so_run_cache", "_dkd_run_cache", "_zero_s", "_along_cache", "_plan_store", "_lattice_store", "_hooks")

    def __getstate__(self):
        """State for `copy.deepcopy`, `pickle` and `torch.save`: everything but the derived caches."""
        state = self.__dict__.copy()
        for key in self._DERIVED_STATE:
            if key in state:
                state[key] = None
        return state

    def _apply(self, fn, recurse=True):
        # .to() / .double() / .cuda() replace the buffers without going through __setattr__
        out = super()._apply(fn, recurse)
        if "_revision" in self.__dict__:
            self._touch()
        return out

    def _touch(self, soft: bool = False) -> None:
        """An attribute of this element was (re)assigned: drop its cached maps, move its revision and the process-wide
        epoch (`Segment` re-validates a run's persistent device plan only when the epoch moved — an O(1) check per track).
        `soft`: nothing but the address of a setting changed (see `_hard_epoch`)."""
        self.__dict__["_revision"] += 1
        self.__dict__["_map_cache"] = None
        self.__dict__["_tmap_cache"] = None
        Element._epoch += 1
        if not soft:
            Element._hard_epoch = Element._epoch

    def register_buffer_or_parameter(self, name: str, value, persistent: bool = True) -> None:
        if isinstance(value, nn.Parameter):
            self.register_parameter(name, value)
        else:
            self.register_buffer(name, value, persistent)

    @property
    def defining_features(self) -> list[str]:
        return ["name"] if len(self.supported_tracking_methods) == 1 else ["name", "tracking_method"]

    @property
    def defining_tensors(self) -> list[str]:
        return [f for f in self.defining_features if isinstance(getattr(self, f), torch.Tensor)]

    def clone(self) -> "Element":
        return self.__class__(
            **{f: (getattr(self, f).clone() if isinstance(getattr(self, f), torch.Tensor)
                   else deepcopy(getattr(self, f))) for f in self.defining_features},
            metadata=deepcopy(self.metadata), sanitize_name=False)

    def plot(self, *args, **kwargs):
        """element.py:318-331 — drawing is outside this tracking engine."""
        raise NotImplementedError(f"{type(self).__name__}.plot: plotting and 3-D meshes are outside this tracking engine; "
                                  "write the lattice with `to_lattice_json` and draw it with the reference")

    def to_mesh(self, *args, **kwargs):
        """element.py:333-362 — see `plot`."""
        raise NotImplementedError(f"{type(self).__name__}.to_mesh: plotting and 3-D meshes are outside this tracking engine; "
                                  "write the lattice with `to_lattice_json` and draw it with the reference")

    def split(self, resolution: torch.Tensor) -> list["Element"]:
        """Slices no longer than `resolution` (element.py:338-347); elements that cannot be split return [self]."""
        return [self]

    def _split_evenly(self, resolution) -> list["Element"]:
        """`num_splits = ceil(max|length| / resolution)` equal slices, everything else unchanged (drift.py:160-173,
        quadrupole.py:261-278, solenoid.py:126-141)."""
        n = max(int(torch.ceil(self.length.abs().max() / resolution).item()), 1)
        parts = []
        for i in range(n):
            part = self.clone()
            part.length = self.length / n
            part.name = f"{self.name}_split_{i}"
            part.metadata = self.metadata
            parts.append(part)
        return parts

    #: merge rules (drift.py:175-187, quadrupole.py:280-301, sextupole.py:133-153, solenoid.py:143-157):
    #: attributes that must be equal, attributes averaged with the lengths as weights, attributes summed
    _merge_equal: tuple[str, ...] | None = None
    _merge_weighted: tuple[str, ...] = ()
    _merge_summed: tuple[str, ...] = ()

    def merge(self, other: "Element") -> "Element | None":
        """Merged element, or None when the two cannot be merged / the type does not support it (element.py:349-358)."""
        if self._merge_equal is None or type(self) is not type(other):
            return None
        for attr in self._merge_equal:
            a, b = getattr(self, attr), getattr(other, attr)
            if not (a.equal(b) if isinstance(a, torch.Tensor) else a == b):
                return None
        merged = self.clone()
        total = self.length + other.length
        for attr in self._merge_weighted:
            setattr(merged, attr, (getattr(self, attr) * self.length + getattr(other, attr) * other.length) / total)
        for attr in self._merge_summed:
            setattr(merged, attr, getattr(self, attr) + getattr(other, attr))
        merged.length = total
        merged.name = merge_element_names(self.name, other.name)
        merged.metadata = {**other.metadata, **self.metadata}
        return merged

    def __repr__(self) -> str:
        feats = ", ".join(f"{f}={getattr(self, f)!r}" for f in self.defining_features)
        return f"{self.__class__.__name__}({feats})"

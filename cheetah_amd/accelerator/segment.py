"""Segment (mirror of cheetah/accelerator/segment.py:45-71, 525-574, 658-700 for the tracking path).

`track` follows the reference's orchestration: an all-skippable lattice is composed into ONE 7x7 map
(`chx_compose_maps`, fp64 accumulation) and applied in one pass over the particles; otherwise the
element list is partitioned into maximal skippable runs (each composed and applied once) and the
non-skippable elements (active Cavity / Screen / SpaceChargeKick / BPM / Aperture) are tracked one by
one.

Host-side cost matters here (the merged pass is a ~10 us kernel), so unlike the reference
 * no nn.Module sub-segments are built per call (segment.py:558-569);
 * the partition into runs is planned once per lattice *revision* (any `setattr` on an element bumps
   its revision counter) and re-used;
 * per run, the composed map, the stacked per-element map table and the summed length are cached
   against a token made of the `_version` counters of the run's tensors (in-place updates), the
   `requires_grad` flags of its parameters and the identity/version of the beam energy.

 * a run whose elements all take device scalars (the usual control loop) keeps a persistent device plan (`_FastRun`,
   `chx_run_track`): the device itself checks whether any setting changed since the stored map was composed, the host
   only compares one process-wide epoch counter that every attribute assignment on an element moves. One C call per
   run and track, no per-tensor Python work.

`track_elementwise` is the merge-free variant (`for e in elements: beam = e.track(beam)`): the E maps
are applied back to back by `chx_track_elementwise` / `chx_track_fused` from a single C call.
"""

from __future__ import annotations

import ctypes
import os
import weakref
from copy import deepcopy

import torch
from torch import nn

from .. import _lib, _ops
from ..particles.parameter_beam import ParameterBeam
from ..particles.particle_beam import ParticleBeam
from ..particles.species import Species
from . import _planner
from ._plans import _IDENTITY, _FastRun, _LatticePlan, _Run  # noqa: F401  (the plan classes; tests reach them through this module)
from .element import Element, tracking_call
from .space_charge_kick import SpaceChargeKick


def _any_requires_grad_py(*tensors) -> bool:
    for t in tensors:
        if t.requires_grad:
            return True
    return False


#: one C call over the tensors of a run: `cheetah_amd._chxtorch` (csrc/chx_torch_host.cpp) reads the flag straight from the
#: tensor objects — 0.9 us for the 300 setting tensors of the 100-element FODO where torch._C._any_requires_grad's argument
#: parser takes 7-13 us; without the extension (a torch upgrade without a rebuild) torch's own function, then the Python loop
try:
    from .._chxtorch import any_requires_grad as _scan_requires_grad

    def _any_requires_grad(*tensors) -> bool:
        return _scan_requires_grad(tensors)
except ImportError:  # pragma: no cover - stale build
    _any_requires_grad = getattr(torch._C, "_any_requires_grad", _any_requires_grad_py)


class _ElementList(nn.ModuleList):
    """The element list of a Segment: any change of the list moves the process-wide epoch (Element._touch), so that plans
    derived from it are rebuilt."""

    def _get_name(self):
        return "ModuleList"      # prints like the reference's element list

    def _moved(self):
        Element._epoch += 1
        Element._hard_epoch = Element._epoch

    def __setitem__(self, idx, module):
        self._moved()
        return super().__setitem__(idx, module)

    def __delitem__(self, idx):
        self._moved()
        return super().__delitem__(idx)

    def __iadd__(self, modules):
        self._moved()
        return super().__iadd__(modules)

    def insert(self, index, module):
        self._moved()
        return super().insert(index, module)

    def append(self, module):
        self._moved()
        return super().append(module)

    def extend(self, modules):
        self._moved()
        return super().extend(modules)

    def pop(self, key=-1):
        self._moved()
        return super().pop(key)


_CHECK_PLANS = os.environ.get("CHX_CHECK_PLANS", "0") == "1"
#: drift-kick-drift kinds chx_dkd_chain carries through a run in registers
_DKD_IN_REGISTERS = (_ops.DKD_KIND["drift"], _ops.DKD_KIND["quadrupole"], _ops.DKD_KIND["dipole"])
#: CHX_SC_CHAIN = auto (default) | on | off — how `Segment.track` takes [SpaceChargeKick, linear run, SpaceChargeKick, ...]:
#: "auto" starts on the tile-ordered chain and lets the asynchronous guard (`_chain_allowed`) send a plan whose beam reshuffles
#: between kicks back to kick-by-kick tracking. The two paths sum the charge in different orders, so WHEN the guard's header
#: arrives decides the last bits of the first few tracks of a plan; "on" / "off" pin the path for bit-reproducible runs.
_CHAIN_MODE = os.environ.get("CHX_SC_CHAIN", "auto").lower()
if _CHAIN_MODE not in ("auto", "on", "off"):
    raise ValueError(f"CHX_SC_CHAIN must be auto, on or off, not {_CHAIN_MODE!r}")


class _HostProxy:
    """`cheetah_amd._chxhost`, loaded (and bound to libchx) at first use."""

    def __getattr__(self, name):
        h = _lib.host()
        global _HOST
        _HOST = h
        return getattr(h, name)


_HOST = _HostProxy()


class _TorchHostProxy:
    """`cheetah_amd._chxtorch`, loaded (and bound to libchx) at first use."""

    def __getattr__(self, name):
        h = _lib.torch_host()
        global _TORCH_HOST
        _TORCH_HOST = h
        return getattr(h, name)


_TORCH_HOST = _TorchHostProxy()


class Segment(Element):
    """Ordered sequence of elements."""

    _static_skippable = False  # depends on the children

    def __init__(self, elements: list[Element], name=None, sanitize_name=None, metadata=None, device=None,
                 dtype=None) -> None:
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, device=device, dtype=dtype)
        del self._buffers["length"]  # `length` is a derived property here (segment.py:54-58)
        self.elements = _ElementList(elements)
        by_name: dict[str, list[Element]] = {}
        for e in elements:
            by_name.setdefault(e.name, []).append(e)
        self.__dict__["_by_name"] = by_name
        self.__dict__["_plan_cache"] = None
        self.__dict__["_chain_guard_state"] = None

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

    # ---- planning ------------------------------------------------------------------------------------
    def _revision_key(self):
        flat = self.__dict__.get("_flat_elements")
        if flat is None or flat[0] != len(self.elements):
            flat = (len(self.elements), [m for m in self.modules() if isinstance(m, Element)])
            self.__dict__["_flat_elements"] = flat
        return tuple([m.__dict__["_revision"] for m in flat[1]])

    def _leaves(self) -> list:
        """The elements in tracking order with plain nested Segments (exactly this class, the stock `track`) replaced by their
        own leaves."""
        out = []
        for e in self.elements:
            if type(e) is Segment and type(e).track is Segment.track:
                out += e._leaves()
            else:
                out.append(e)
        return out

    def _plan(self):
        """[(kind, payload)] with kind 'run' (payload _Run) or 'element' (payload Element). Depends on the element list
        and on which elements are skippable — not on their settings."""
        cached = self.__dict__["_plan_cache"]
        if cached is not None and cached[2] >= Element._hard_epoch and cached[3] is None:
            # nothing but addresses of settings was assigned anywhere since (Element._hard_epoch), and no element's skippability
            # depends on tensor values
            return cached[1]
        # plain nested Segments are planned THROUGH: their elements join the parent's runs, stretches and chains (the reference
        # tracks a nested segment as one element, merged or walked by its own `track` — the same maps in another association).
        # A lattice file's cells-of-cells would otherwise be 25 non-static "elements" walked one by one: 625 us instead of 28
        # for 100 elements.
        elements = self._leaves()
        ids = tuple([id(e) for e in elements])
        revs = [e.__dict__["_revision"] for e in elements]
        if cached is not None and cached[0][0] == ids:
            # the same element objects: only an element whose own revision moved, or one whose skippability depends on
            # tensor VALUES (Cavity voltage, nested segments), can have changed its skippability
            skippable = list(cached[0][1])
            for i, (rev, old) in enumerate(zip(revs, cached[4])):
                if rev != old or not elements[i]._static_skippable:
                    skippable[i] = elements[i].is_skippable
            skippable = tuple(skippable)
        else:
            skippable = tuple([e.is_skippable for e in elements])
        key = (ids, skippable)
        if cached is not None and cached[0] == key:
            self.__dict__["_plan_cache"] = (key, cached[1], Element._epoch, cached[3], revs)
            return cached[1]
        # a partition seen before (a cavity switched off and on again, a diagnostic toggled): its runs come back with their
        # persistent plans, which re-validate against the epoch — rebuilding every plan of a 16-cell linac costs ~0.9 ms
        store = self.__dict__.get("_plan_store")
        if store is None:                    # (never planned, or a copy: derived state does not travel)
            store = self.__dict__["_plan_store"] = {}
        known = store.get(key)
        if known is not None:
            self.__dict__["_plan_cache"] = (key, known[0], Element._epoch, known[1], revs)
            return known[0]
        plan, run = [], []
        for e in elements:
            if e.is_skippable:
                run.append(e)
            else:
                if run:
                    plan.append(("run", _Run(run)))
                    run = []
                plan.append(("element", e))
        if run:
            plan.append(("run", _Run(run)))
        # elements whose skippability is a function of tensor VALUES (Cavity voltage, nested segments): the list must be
        # re-examined on every call while there are any
        dynamic = [m for e in elements for m in e.modules() if isinstance(m, Element) and not m._static_skippable] or None
        self.__dict__["_plan_cache"] = (key, plan, Element._epoch, dynamic, revs)
        if len(store) >= 8:
            store.clear()
            self.__dict__.pop("_lattice_store", None)
        store[key] = (plan, dynamic)
        return plan

    @staticmethod
    def _lead_at(lead_in, acc, energy_shape) -> tuple:
        """Batch shape of a beam at a point of a stretch: what it came in with, spread over the vectorised settings in front of
        that point (`acc`) and over the beam energies once a map was applied."""
        # (plain tuples, right-aligned: `torch.broadcast_shapes` costs ~6 us a call, and a lattice has a monitor per cell)
        out = tuple(lead_in)
        for other in (acc, tuple(energy_shape)):
            if not other or other == out:
                continue
            a, b = ((1,) * (len(other) - len(out)) + out, tuple(other)) if len(other) > len(out) else (out, (1,) * (len(out) - len(other)) + tuple(other))
            out = tuple(x if y == 1 else y for x, y in zip(a, b))
        return out

    def _lattice_cache_for(self, plan):
        """The stretch plans of `plan` (one table per partition of the lattice that `_plan` keeps)."""
        store = self.__dict__.get("_lattice_store")
        if store is None:
            store = self.__dict__["_lattice_store"] = {}
        cache = store.get(id(plan))
        if cache is None or cache[0] is not plan:
            if len(store) >= 8:
                store.clear()
            cache = store[id(plan)] = (plan, {})
        self.__dict__["_lattice_cache"] = cache
        return cache

    # ---- per-run products ------------------------------------------------------------------------------
    @staticmethod
    def _refresh(run: _Run, energy, species) -> bool:
        """Invalidate the run's caches when its token changed. Returns whether caching is allowed."""
        if energy.requires_grad or species.mass_eV.requires_grad or _ops.CAPTURING[0]:
            run.token = run.tm = run.stack = None
            return False
        token = run.current_token(energy, species)
        if token != run.token:
            run.token, run.tm, run.stack, run.s_cache = token, None, None, None
            run.energy_ref = energy  # kept alive so that its id cannot be recycled
        return True

    @staticmethod
    def _run_length(run: _Run):
        """Summed length of the run, redone only when a length tensor was replaced or modified. The previous tensors are
        kept referenced by the key, so that an `is` comparison cannot be fooled by a recycled object id."""
        key = run.length_key
        if key is not None and key[0] == Element._epoch and run.length is not None and not _ops.CAPTURING[0]:
            # no attribute of any element was assigned since the sum was formed and every length is a tensor the element itself
            # holds (not a derived property): the same tensor objects — only an in-place edit can have changed them (100
            # `nn.Module.__getattr__` look-ups, ~20 us, for a 100-element run otherwise)
            if all(t._version == v for t, v in key[1]):
                return run.length
        lengths = [e.length for e in run.elements]
        same = (key is not None and run.length is not None and not run.length.requires_grad and not _ops.CAPTURING[0]
                and len(key[1]) == len(lengths) and all(a is b and a._version == v for a, (b, v) in zip(lengths, key[1])))
        plain = all(("length" in e._buffers) or ("length" in e._parameters) for e in run.elements)
        if not same:
            total = None
            for t in lengths:
                total = t if total is None else total + t
            run.length = total
            # derived lengths (a sub-segment's sum is a fresh tensor every time) never compare identical: recomputed
            run.s_cache = None
        # (the epoch shortcut above only for sums without a graph over lengths the elements hold themselves)
        run.length_key = (Element._epoch if (plain and not run.length.requires_grad) else None, [(t, t._version) for t in lengths])
        return run.length

    @staticmethod
    def _run_s(run: _Run, s_in: torch.Tensor) -> torch.Tensor:
        """`incoming.s + length` of the run; the sum is reused while the same incoming `s` tensor is
        tracked again (the usual RL loop re-tracks one incoming beam), saving a device op per track."""
        length = Segment._run_length(run)      # re-validated on every call (and resets s_cache when a length changed)
        c = run.s_cache
        if c is not None and c[0] is s_in and c[1] == s_in._version and not s_in.requires_grad and not _ops.CAPTURING[0]:
            return c[2]
        s_out = s_in + length
        if not s_out.requires_grad:
            run.s_cache = (s_in, s_in._version, s_out)
        return s_out

    @staticmethod
    def _run_map_grad(run: _Run, energy, species):
        """The run's composed map WITH a graph from the persistent differentiable plan (`_ops.RunMapPlanned`: one C call
        forward, one backward), or None when nothing requires grad / the run does not qualify (vectorised settings or energy,
        a species whose mass carries a gradient)."""
        if energy.dim() != 0 or not energy.is_cuda or species.mass_eV.requires_grad \
                or species.num_elementary_charges.requires_grad:
            return None
        fr = run.gfast
        if fr is None or fr.dtype != energy.dtype or fr.device != energy.device:
            fr = run.gfast = _FastRun(run, energy.dtype, energy.device, allow_grad=True)
        elif fr.epoch != Element._epoch:
            fr.refresh()
        if not fr.ok or not (energy.requires_grad or _any_requires_grad(*fr.distinct)):
            return None
        if _CHECK_PLANS:
            fr.verify()
        return _ops.RunMapPlanned.apply(fr, energy, species.mass_eV_float, species.num_elementary_charges_float, *fr.distinct)

    @staticmethod
    def _run_map(run: _Run, energy, species) -> torch.Tensor:
        if torch.is_grad_enabled():
            tm = Segment._run_map_grad(run, energy, species)
            if tm is not None:
                return tm
        cacheable = Segment._refresh(run, energy, species)
        if cacheable and run.tm is not None:
            return run.tm
        tm = None
        if not (torch.is_grad_enabled() and (energy.requires_grad or species.mass_eV.requires_grad)):
            tm = Segment._run_map_vector(run, energy, species)
        if tm is None and (cacheable or not species.mass_eV.requires_grad):
            # all-scalar runs (the usual control loop, and gradient-based tuning of scalar settings): every element's map and
            # their product in two C calls — with gradients ONE autograd node for the run (_ops.RunMapScalars)
            tm = _ops.build_compose_scalars(run.elements, energy, species.mass_eV_float, species.num_elementary_charges_float)
        if tm is None:
            maps = [e.first_order_transfer_map(energy, species) for e in run.elements]
            batch_shape = torch.broadcast_shapes(energy.shape, *[m.shape[:-2] for m in maps])
            tm = _ops.compose_maps(maps, batch_shape, maps[0].dtype, maps[0].device)
        if cacheable and not tm.requires_grad:
            run.tm = tm
        return tm

    @staticmethod
    def _vector_run_rows(run: _Run, dtype, device, common=None):
        """(kinds, per-element setting addresses, per-element flags (1 = a tensor of the batch shape), tensors, batch shape, expanded,
        own shape) of a run whose elements all have a device builder and settings that are device scalars or contiguous tensors that
        BROADCAST to one batch shape (`common` when given — the shape of the whole stretch) — or None (a vectorised length, shapes
        that do not broadcast, another dtype / device). A setting whose own shape is not the batch shape — (8, 1) against (1, 8) in
        a grid scan — is addressed through an expanded contiguous COPY: `expanded` = [(setting, copy, [version])], refreshed by the
        user of the tables when the setting's version moved (`_refresh_expanded`). `own shape`: what the run's own settings
        broadcast to (the batch shape a beam has BEHIND this run in the walk). Whether a setting requires grad is asked by the
        caller, per call (`_any_requires_grad(*tensors)`)."""
        kinds, refs_all, shapes = [], [], []
        for e in run.elements:
            kind = getattr(e, "_chx_kind", None)
            if kind is None or not e._plannable():
                return None
            if kind == _IDENTITY:
                continue
            refs = e._builder_scalar_refs()
            for k, (t, index) in enumerate(refs):
                if t.dtype != dtype or t.device != device:
                    return None
                if index is not None:                      # a component of a vector-valued setting (misalignment[..., i])
                    if t.dim() != 1 or not t.is_contiguous():
                        return None
                elif t.dim() != 0:
                    if k == 0 or not t.is_contiguous():
                        return None
                    shapes.append(tuple(t.shape))
            kinds.append(kind)
            refs_all.append(refs)
        shape = own = None
        if shapes:
            try:
                own = tuple(torch.broadcast_shapes(*shapes))
                shape = tuple(torch.broadcast_shapes(own, tuple(common))) if common is not None else own
            except RuntimeError:
                return None
            if common is not None and shape != tuple(common):
                return None
        rows, flags, keep, expanded = [], [], [], []
        for refs in refs_all:
            row, fl = [], []
            for t, index in refs:
                if index is not None:
                    row.append(t.data_ptr() + index * t.element_size())
                    fl.append(0)
                elif t.dim() == 0:
                    row.append(t.data_ptr())
                    fl.append(0)
                else:
                    if tuple(t.shape) != shape:
                        with torch.no_grad():
                            copy = t.expand(shape).contiguous()
                        expanded.append((t, copy, [t._version]))
                        keep.append(copy)
                        row.append(copy.data_ptr())
                    else:
                        row.append(t.data_ptr())
                    fl.append(1)
                keep.append(t)
            rows.append(row)
            flags.append(fl)
        return kinds, rows, flags, keep, shape, expanded, own

    @staticmethod
    def _refresh_expanded(expanded) -> None:
        """The expanded copies of settings whose own shape is not the batch shape follow in-place edits of the settings. While a
        device graph records, the copy is made unconditionally: it becomes a node of the graph, so a replay follows settings that
        were written between two replays (a version counter is host state the replay does not see)."""
        recording = torch.cuda.is_current_stream_capturing()
        for src, copy, version in expanded:
            if recording or src._version != version[0]:
                with torch.no_grad():
                    copy.copy_(src.expand(copy.shape))
                version[0] = src._version

    @staticmethod
    def _vector_tables(run: _Run, dtype, device):
        """(E, kinds, addresses, flags, tensors, batch shape, expanded copies) of `_vector_run_rows` packed for `chx_run_map_batched`, kept while the
        epoch stands still — or None."""
        c = run.vrows
        if c is None or c[0] != Element._epoch or c[1] != dtype or c[2] != device:
            if torch.cuda.is_current_stream_capturing():
                return None       # (tables — and expanded copies — are not built inside a recording; the general path is capturable)
            got = Segment._vector_run_rows(run, dtype, device) if len(run.elements) <= 192 else None
            if got is not None:
                kinds, rows, row_flags, keep, shape, expanded, _ = got
                ptrs, flags = [], []
                for r, f in zip(rows, row_flags):
                    ptrs += r + [None] * (_ops.MAX_PARAMS - len(r))
                    flags += f + [0] * (_ops.MAX_PARAMS - len(f))
                E = len(kinds)
                got = (E, (ctypes.c_int32 * E)(*kinds), (ctypes.c_void_p * (E * _ops.MAX_PARAMS))(*ptrs),
                       (ctypes.c_uint8 * (E * _ops.MAX_PARAMS))(*flags), tuple(keep), shape, tuple(expanded))
            c = run.vrows = (Element._epoch, dtype, device, got)
        return c[3]

    @staticmethod
    def _run_map_vector(run: _Run, energy, species):
        """The composed maps (*shape, 7, 7) of a run whose settings are vectorised over a batch of lattice settings — some
        parameters tensors of ONE common shape, the others scalars; scalar energy and lengths — by one launch
        (`chx_run_map_batched`: a workgroup per batch row builds and composes the row's maps), or None when the run does not
        qualify (no vectorised setting at all, mixed shapes, gradients, a vectorised length or vector component). The general
        path builds every vectorised element's maps on its own: ~100 us of host time per element and step, whatever the batch."""
        if not energy.is_cuda or len(run.elements) > 192 or (energy.dim() != 0 and not energy.is_contiguous()):
            return None
        # the packed tables stand while no attribute of any element was assigned (the addresses of the settings; their VALUES are
        # read by the device on every call): building them anew is ~3 us per element and step
        tables = Segment._vector_tables(run, energy.dtype, energy.device)
        if tables is None:
            return None
        E, kinds_arr, ptrs_arr, flags_arr, keep, shape, expanded = tables
        if torch.is_grad_enabled() and _any_requires_grad(*keep):
            return None
        if expanded:
            Segment._refresh_expanded(expanded)
        if energy.dim() != 0:
            # a scan of BEAM ENERGIES (with scalar settings, or settings vectorised over the same shape): row b's maps are built
            # for energy b. Element by element that is one builder call per element and step — every cavity in front hands on a
            # new energy tensor, so nothing is ever cached: 48 elements of a linac 4.7 ms
            if shape is not None and tuple(energy.shape) != tuple(shape):
                return None
            shape = tuple(energy.shape)
        dtype, device = energy.dtype, energy.device
        if shape is None or E == 0:
            return None
        B = _ops.numel(shape)
        lib = _lib.lib()
        code = _ops.dtype_code(dtype)
        ws_bytes = lib.chx_run_map_batched_workspace_bytes(E, B, code)
        ws = _ops.workspace(ws_bytes, device) if ws_bytes else None
        R = torch.empty((*shape, 7, 7), dtype=dtype, device=device)
        _ops.check(lib.chx_run_map_batched(kinds_arr, ptrs_arr, flags_arr, E, B, energy.data_ptr(),
                                           1 if energy.dim() != 0 else 0, species.mass_eV_float, species.num_elementary_charges_float, code,
                                           ws.data_ptr() if ws is not None else None, ws_bytes, R.data_ptr(), _ops.stream_ptr()),
                   "chx_run_map_batched")
        return R

    @staticmethod
    def _run_stack(run: _Run, energy, species) -> torch.Tensor:
        cacheable = Segment._refresh(run, energy, species)
        if cacheable and run.stack is not None:
            return run.stack
        maps = []
        for e in run.elements:
            m = e.first_order_transfer_map(energy, species)
            maps.append(m)
        bshape = torch.broadcast_shapes(energy.shape, *[m.shape[:-2] for m in maps])
        Bm = _ops.numel(bshape)
        stack = torch.stack([m.expand(*bshape, 7, 7).reshape(Bm, 7, 7) for m in maps])
        if cacheable and not stack.requires_grad:
            run.stack = stack
        return stack

    @staticmethod
    def _run_apply_fast(run: _Run, incoming: ParticleBeam):
        """Tracked particles of `incoming` through `run` by ONE C call (chx_run_track), or None when the run or the beam
        does not qualify (vectorised / trainable settings, a vectorised beam, gradients) and the general path is taken."""
        p = incoming.particles
        if p.dim() < 2 or not p.is_cuda or (p.dim() > 2 and not p.is_contiguous()):
            return None                    # (B beams in one ParticleBeam under scalar settings: one flat beam of B N particles)
        fr = run.fast
        if fr is None or fr.dtype != p.dtype or fr.device != p.device:
            fr = run.fast = _FastRun(run, p.dtype, p.device)
        elif fr.epoch != Element._epoch:
            fr.refresh()
        if not fr.ok:
            return None
        if _CHECK_PLANS:
            fr.verify()
        e = incoming.energy
        if e.dim() != 0 or e.dtype != fr.dtype or e.device != fr.device:
            return None
        sp = incoming.species
        if torch.is_grad_enabled():
            if p.requires_grad or e.requires_grad or sp.mass_eV.requires_grad or sp.num_elementary_charges.requires_grad:
                return None
            if _any_requires_grad(*fr.tensors):   # a buffer switched with requires_grad_(True) in place moves no counter
                return None
        x = p if p.is_contiguous() and p.data_ptr() % 16 == 0 else _ops.aligned(p)
        if x.dim() > 2:
            x = x.reshape(-1, 7)
        _ops.check_current_device(fr.device)
        s_in = incoming.s
        on_device = s_in.dim() == 0 and s_in.dtype == fr.dtype and s_in.device == fr.device and not s_in.requires_grad
        # the C host step (csrc/chx_host.c): output tensors, stream, chx_run_track — one call, no ctypes marshalling. The path
        # length comes from the same launch that validates the settings (no host copy of the lengths to go stale)
        out, s_out = _HOST.run_track(fr.capsule, x, x.shape[0], e, s_in if on_device else None, sp.mass_eV_float,
                                     sp.num_elementary_charges_float, fr.device.index)
        if p.dim() > 2:
            out = out.reshape(p.shape)
        return out, (s_out if s_out is not None else Segment._run_s(run, s_in))

    @staticmethod
    def _run_map_fast(run: _Run, ref: torch.Tensor, energy: torch.Tensor, species: Species, s_in: torch.Tensor):
        """(map, s_out): the run's composed map as a (7, 7) tensor living inside the persistent plan's device state (chx_run_map:
        one launch that re-validates the settings and rebuilds only if one changed) and the path length behind the run from
        the same launch — or None when the run does not qualify. `ref` gives dtype and device. The map tensor is overwritten
        by the next call: for immediate use on the same stream."""
        if not ref.is_cuda or energy.dim() != 0:
            return None
        fr = run.fast
        if fr is None or fr.dtype != ref.dtype or fr.device != ref.device:
            fr = run.fast = _FastRun(run, ref.dtype, ref.device)
        elif fr.epoch != Element._epoch:
            fr.refresh()
        if not fr.ok or energy.dtype != fr.dtype or energy.device != fr.device:
            return None
        if torch.is_grad_enabled() and (energy.requires_grad or species.mass_eV.requires_grad
                                        or species.num_elementary_charges.requires_grad or _any_requires_grad(*fr.tensors)):
            return None
        R_addr = ctypes.c_void_p()
        s_out = Segment._device_s(fr, s_in)
        _ops.check(_lib.lib().chx_run_map(fr.kinds, fr.ptrs, fr.E, energy.data_ptr(), species.mass_eV_float,
                                          species.num_elementary_charges_float, fr.code, fr.state.data_ptr(), fr.state_bytes,
                                          ctypes.byref(R_addr), s_in.data_ptr() if s_out is not None else None,
                                          s_out.data_ptr() if s_out is not None else None, _ops.stream_ptr()), "chx_run_map")
        view = fr.R_view
        if view is None or view.data_ptr() != R_addr.value:
            off = R_addr.value - fr.state.data_ptr()
            view = fr.R_view = fr.state.view(torch.uint8)[off:off + 49 * ref.element_size()].view(ref.dtype).view(7, 7)
        return view, (s_out if s_out is not None else Segment._run_s(run, s_in))

    @staticmethod
    def _device_s(fr: _FastRun, s_in: torch.Tensor):
        """A fresh scalar for the path length behind the run when the plan's launch can write it (one value of the plan's
        dtype on its device, no graph), else None (the caller adds the validated lengths on the host side)."""
        if s_in.dim() == 0 and s_in.dtype == fr.dtype and s_in.device == fr.device and not s_in.requires_grad:
            return torch.empty_like(s_in)
        return None

    #: active Screens as items of the stretch call (False: a screen ends the stretch and is tracked on its own — the walk the
    #: tests compare the stretch with)
    _STRETCH_SCREENS = True

    #: rows of vectorised lattice settings a stretch call takes when a WORKGROUP per (item, row) prepares the maps — i.e. when a run
    #: of the stretch holds more than 64 elements (`_LatticePlan.small_runs == 0`); above: the walk item by item, whose cost does not
    #: depend on the rows (benchmarks/response_matrix_probe.py: 64 rows 1.4 -> 0.14 ms, 4096 rows 1.4 -> 6.4 ms). With runs of at most
    #: 64 elements a WAVE per (item, row) prepares the stretch — bit 0 of `small_runs`: no cavity in it; bit 3
    #: (CHX_LATTICE_SHORT_RUNS, since round 5): with cavities as well, each wave walking the energy through the cavities in front of
    #: its item — and any number of rows up to 65 535 goes: the guards below test `not lp.small_runs` on purpose (either bit lifts
    #: the cap; tests/test_gpu_lattice_stretch.py holds a 4096-row cavity scan to it)
    _STRETCH_MAX_ROWS = 512

    #: a persistent device plan (`_FastRun`) holds at most 192 elements and 400 setting tensors; a longer run is cut into pieces
    _PART_ELEMENTS = 128
    _PART_TENSORS = 380
    #: runs shorter than this keep the general path when their one plan declines (its host walk is a few microseconds)
    _PART_MIN_RUN = 16

    @staticmethod
    def _run_map_parts(run: _Run, ref: torch.Tensor, energy: torch.Tensor, species: Species, s_in: torch.Tensor):
        """(map, s_out) of a run that ONE persistent device plan does not take — a beamline of a thousand elements between two
        screens, or a run with a CustomTransferMap / one vectorised or trainable magnet in it — from its PIECES: stretches of
        elements a plan takes (at most 128 elements / 380 setting tensors each: their composed map comes from `chx_run_map`, the
        device re-validates the piece's settings and rebuilds its map only when one changed) and, one by one, the elements a plan
        cannot take (their map through the general path, cached per element). The K piece maps are composed by one more launch.
        Host work per track: K calls instead of a walk over every element's tensors (~0.85 us per element: 0.85 ms for 1000
        elements, 3.4 ms for 5000; 100 us for 100 elements around one CustomTransferMap). None when that would not pay."""
        if len(run.elements) < Segment._PART_MIN_RUN or not ref.is_cuda or energy.dim() != 0:
            return None
        if len(run.elements) <= 192:
            tables = Segment._vector_tables(run, ref.dtype, ref.device)
            if tables is not None and tables[5] is not None and tables[0] > 0:
                # settings vectorised over ONE batch shape: `_run_map` has all rows' maps from one launch (`_run_map_vector`), and
                # keeps them while no setting changes — the pieces would cost a call each
                return None
        built = run.parts
        if built is None or built[0] is None:
            # the partition: elements whose settings are device scalars without a graph go into the plans' stretches. It is redone
            # (at most once per epoch) when a stretch's plan declines after all — a setting became vectorised or trainable since
            parts, piece, refs = [], [], 0
            dtype, device = ref.dtype, ref.device
            for e in run.elements:
                takes = getattr(e, "_chx_kind", None) is not None and e._plannable() and not e._parameters
                settings = e._builder_scalar_refs() if takes else ()
                if takes and any(t.dtype != dtype or t.device != device or t.requires_grad or t.dim() != (0 if index is None else 1)
                                 for t, index in settings):
                    takes = False
                if not takes:
                    if piece:
                        parts.append((_Run(piece), True))
                        piece, refs = [], 0
                    parts.append((_Run([e]), False))
                    continue
                n = len(settings)
                if piece and (len(piece) >= Segment._PART_ELEMENTS or refs + n > Segment._PART_TENSORS):
                    parts.append((_Run(piece), True))
                    piece, refs = [], 0
                piece.append(e)
                refs += n
            if piece:
                parts.append((_Run(piece), True))
            if len(parts) < 2 or not any(cand and len(part.elements) >= 8 for part, cand in parts):
                parts = ()
            built = run.parts = (parts, Element._epoch if built is None else built[1])
        parts = built[0]
        if not parts:
            return None
        maps, s, views = [], s_in, []
        for part, candidate in parts:
            got = Segment._run_map_fast(part, ref, energy, species, s) if candidate else None
            if got is None:             # (a vectorised or gradient-carrying setting: this piece through the general path)
                if candidate and built[1] != Element._epoch:
                    run.parts = (None, Element._epoch)       # partition again on the next track (once per epoch)
                maps.append(Segment._run_map(part, energy, species))
                s = Segment._run_s(part, s)
            else:
                views.append(len(maps))
                maps.append(got[0])
                s = got[1]
        if torch.is_grad_enabled() and any(m.requires_grad for m in maps):
            # `ComposeMaps` saves its inputs for the backward pass: a piece map that is a VIEW of its plan's device state would be
            # overwritten by the next forward pass (another beam energy, an in-place edit of a setting) without any version
            # counter moving — the product's backward pass needs the values of THIS forward pass
            for k in views:
                maps[k] = maps[k].clone()
        batch_shape = torch.broadcast_shapes(*[m.shape[:-2] for m in maps])
        return _ops.compose_maps(maps, batch_shape, ref.dtype, ref.device), s

    def first_order_transfer_map(self, energy: torch.Tensor, species: Species):
        plan = self._plan()
        if len(plan) == 1 and plan[0][0] == "run":
            return self._run_map(plan[0][1], energy, species)
        if not plan:
            return torch.eye(7, dtype=energy.dtype, device=energy.device).repeat(*energy.shape, 1, 1)
        return None

    # ---- tracking ---------------------------------------------------------------------------------------
    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        from .marker import _unaliased

        # the walk hands tensors from element to element without copies; if nothing on the way produced new coordinates
        # (markers, inactive diagnostics only) the result is copied once here, like the reference's `incoming.clone()`
        return _unaliased(self._track_internal(incoming), incoming)

    @tracking_call
    def _track_internal(self, incoming: ParticleBeam) -> ParticleBeam:
        if isinstance(incoming, ParameterBeam):
            plan = self._plan()
            i, n_items = 0, len(plan)
            while i < n_items:
                kind, item = plan[i]
                i += 1
                if (kind == "run" or item._is_cavity or item._is_bpm or item._is_screen) and n_items - i >= 1:
                    done = self._lattice_stretch_parameter(plan, i - 1, incoming)
                    if done is not None:
                        incoming, i = done
                        continue
                if kind == "run":
                    fast = None
                    if not (torch.is_grad_enabled() and (incoming.mu.requires_grad or incoming.cov.requires_grad)):
                        fast = self._run_map_fast(item, incoming.mu, incoming.energy, incoming.species, incoming.s)
                        if fast is None and len(item.elements) >= self._PART_MIN_RUN:
                            fast = self._run_map_parts(item, incoming.mu, incoming.energy, incoming.species, incoming.s)
                    if fast is None:
                        tm, s_out = self._run_map(item, incoming.energy, incoming.species), self._run_s(item, incoming.s)
                    else:
                        tm, s_out = fast
                    mu, cov = _ops.parameter_track(incoming.mu, incoming.cov, tm)
                    incoming = ParameterBeam(mu, cov, incoming.energy, total_charge=incoming.total_charge, s=s_out,
                                             species=incoming.species)
                else:
                    incoming = item._track_internal(incoming)
            return incoming
        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        # every plan item through the first path of the planner's table that takes it (accelerator/_planner.py: the space-charge
        # chain, the stretch call, runs ahead of / made of non-linear elements, the merged run, kick + run, the element's own track)
        return _planner.walk_particles(self, self._plan(), incoming)

    def _lattice_stretch(self, plan, i: int, incoming: ParticleBeam, allow_screens: bool = True):
        """plan[i] and the items behind it as ONE `chx_lattice_track` call when they form a stretch [run | active Cavity]+ (at
        least two items, at least one cavity) of scalar settings and the beam is one plain beam without a graph: (outgoing beam,
        index behind the stretch), else None. Same numbers, bit for bit, as the walk item by item."""
        cache = self.__dict__.get("_lattice_cache")
        if cache is None or cache[0] is not plan:
            cache = self._lattice_cache_for(plan)
        p = incoming.particles
        e = incoming.energy
        # one energy: active Screens are items of the stretch too (their record and image come from the same two launches) — for one
        # plain beam and for a vectorised one (B beams under the one lattice setting: every record and image holds B of them). Where
        # the screens' host step does not apply after all (vectorised settings, charges per beam, ...) the stretch is taken again
        # without them: it then ends in front of the first screen (`allow_screens` False)
        with_screens = allow_screens and Segment._STRETCH_SCREENS and p.dim() >= 2 and e.dim() == 0
        key = (i, p.dtype, p.device, with_screens)
        entry = cache[1].get(key)
        if entry is None:
            j, cavities = i, 0
            while j < len(plan) and (plan[j][0] == "run" or plan[j][1]._is_cavity or plan[j][1]._is_bpm or plan[j][1]._is_aperture
                                     or (with_screens and plan[j][1]._is_screen)):
                cavities += plan[j][0] != "run"          # (cavities, active BPMs, apertures, screens: what makes a stretch worth one call)
                j += 1
            entry = cache[1][key] = False if (j - i < 2 or cavities == 0 or not p.is_cuda) else [j, None]
        if entry is False:
            return None
        if p.dim() < 2 or not p.is_cuda or (p.dim() > 2 and not p.is_contiguous()):
            return None                    # (B beams in one ParticleBeam: blockIdx.y of the particle pass)
        s_in, sp = incoming.s, incoming.species
        if e.dtype != p.dtype or e.device != p.device or (e.dim() != 0 and not e.is_contiguous()):
            return None
        energy_rows = e.dim() != 0          # a scan of beam energies: row b of the maps is built for energy b
        lp = entry[1]
        if lp is None or lp.epoch != Element._epoch:
            if torch.cuda.is_current_stream_capturing():
                return None     # (the table's upload is not part of a recording: the walk item by item is capturable as it is)
            if lp is None:
                lp = entry[1] = _LatticePlan(plan[i:entry[0]], p.dtype, p.device, allow_vector=True, allow_screens=with_screens)
            else:
                lp.refresh()
        if lp.edge_keys:
            # a histogram screen's bin edges sit in the table as arrays formed from its pixel size: an in-place edit of that tensor
            # moves no epoch — re-derive the stretch (the cloud-in-cell extent is derived on the device and follows by itself)
            for screen, ps, version in lp.edge_keys:
                if ps._version != version:
                    if torch.cuda.is_current_stream_capturing():
                        return None
                    lp.patch = None
                    lp.refresh()
                    break
        if not lp.ok:
            return None
        if _CHECK_PLANS:
            lp.verify()
        grad_run = None
        if torch.is_grad_enabled() and (p.requires_grad or e.requires_grad or sp.mass_eV.requires_grad
                                        or sp.num_elementary_charges.requires_grad or _any_requires_grad(*lp.tensors)):
            # gradients: [run of scalar settings | one active Screen] on a beam without a graph is ONE differentiable node
            # (cheetah_amd._chxtorch RunScreenTrack); anything else takes the general differentiable path
            grad_run = self._stretch_grad_run(lp, incoming)
            if grad_run is None:
                return None
        x = p if p.is_contiguous() and p.data_ptr() % 16 == 0 else _ops.aligned(p)
        _ops.check_current_device(lp.device)
        on_device = s_in.dim() == 0 and s_in.dtype == p.dtype and s_in.device == p.device and not s_in.requires_grad
        if lp.bpms or lp.screens:
            from .. import sharding

            if sharding.active_group() is not None:
                return None            # a particle-sharded beam: the monitors read GLOBAL means (BPM._track_internal exchanges them),
                #                        a screen sums its image over the ranks
        w_out = incoming.survival_probabilities
        lead_x, N = tuple(p.shape[:-2]), p.shape[-2]
        lead, Bm = lead_x, 1
        if lp.vshape is not None or energy_rows:
            # settings vectorised over a scan of the lattice (and / or a scan of beam energies): row b of the outgoing beams = the
            # beam (ONE shared beam, or its own row b) through row b of the settings at energy b — the preparation launch forms
            # the maps of every row, the particle pass picks its row's by blockIdx.y
            try:
                lead = tuple(torch.broadcast_shapes(lead_x, lp.vshape if lp.vshape is not None else (), e.shape))
            except RuntimeError:
                return None
            if (lp.vshape is not None and tuple(lp.vshape) != lead) or (energy_rows and tuple(e.shape) != lead) \
                    or (_ops.numel(lead_x) != 1 and lead_x != lead):
                return None
            Bm = _ops.numel(lead)
            if Bm > 1 and not lp.small_runs and Bm > Segment._STRETCH_MAX_ROWS:
                return None       # (a workgroup per (item, row) prepares a stretch with runs of more than 64 elements: beyond a few hundred rows the walk)
            if not lp.ensure_rows(Bm):
                return None
        B, Bx = _ops.numel(lead), _ops.numel(lead_x)
        if B < 1 or B > 65535:
            return None
        flags = lp.small_runs | (2 if energy_rows else 0) | (4 if lp.e_out_rows else 0)
        if lp.expanded:
            Segment._refresh_expanded(lp.expanded)
        w_lead = tuple(incoming.survival_probabilities.shape[:-1])
        if lp.screens:
            # [run | cavity | monitor | aperture | active Screen]+ on one plain beam: the C++ host step (cheetah_amd._chxtorch)
            # allocates the outgoing beam, every screen's record and image and enqueues the two launches
            q, w = incoming.particle_charges, incoming.survival_probabilities
            if Bm != 1 or not on_device or q.shape != (N,) or (w.shape != (N,) and tuple(w.shape) != lead + (N,)) or q.dtype != p.dtype \
                    or w.dtype != p.dtype or q.device != p.device or w.device != p.device or not q.is_contiguous() or not w.is_contiguous() \
                    or (torch.is_grad_enabled() and (q.requires_grad or w.requires_grad)) or (lead and grad_run is not None) \
                    or (lead and any(scr.method != "cloud-in-cell" for scr in lp.screens)):      # (screen.py:292-294: the others refuse vectorised beams)
                # (settings, charges or weights the screens' host step does not take: the stretch without its screens)
                return self._lattice_stretch(plan, i, incoming, allow_screens=False)
            if grad_run is not None:
                run, fr = grad_run
                out, rows, C, q_at, w_at, e_at, s_at, sums, maps = _TORCH_HOST.run_screen_track(
                    lp.capsule_s, x, e, s_in, q, w, fr.distinct, fr.grad_meta, sp.mass_eV_float, sp.num_elementary_charges_float)
                x1, origin = x.reshape(1, N, 7), _ops._origin(p)
                # the recorded survival probabilities are the incoming ones (no aperture in a [run | Screen] stretch): tagged as
                # their copy, so that the incoming beam's memoised moments are found again on the next step of a loop; the one-pass
                # sums of the recorded rows ride on the rows (a beam property of them: one small launch, _ops.moment_entry)
                _ops.mark_copy_of(w_at, w)
                rows._chx_partials = (sums, rows._version, w_at,
                                      (C, e, fr.distinct, fr.grad_meta, sp.mass_eV_float, sp.num_elementary_charges_float, maps))
                out._chx_lin = _ops._LinearSource(origin, x1, C, (), out._version)
                lp.screens[0]._record_stretch((rows, q_at, w_at, e_at, s_at, x1, C, origin), N, sp, None, "particles_grad")
                return ParticleBeam(out, e, particle_charges=q, survival_probabilities=w, s=self._run_s(run, s_in), species=sp), i + lp.count
            for ap in lp.apertures:
                ap._check_limits()
            n_bpm = len(lp.bpms)
            readings = ws = None
            ws_bytes = 0
            if n_bpm:
                readings = torch.empty((n_bpm, B, 2), dtype=p.dtype, device=p.device)
                ws_bytes = _lib.lib().chx_lattice_diag_workspace_bytes(N, B, n_bpm)
                ws = _ops.workspace(ws_bytes, p.device)
            if lp.apertures:
                w_out = torch.empty(lead + (N,), dtype=p.dtype, device=p.device)
            from .screen import Screen

            # (a vectorised beam goes in as (B, N, 7): the beams of a (2, 3, N, 7) array one behind the other)
            out, e_out, s_out, records, images = _TORCH_HOST.lattice_track_screens(
                lp.capsule_s, x.reshape(B, N, 7) if lead else x, e, s_in, q, w, sp.mass_eV_float, sp.num_elementary_charges_float,
                lp.device.index, Screen._EAGER_IMAGE_PARTICLES, w_out if lp.apertures else None, n_bpm, readings, ws, ws_bytes)
            for k, bpm in enumerate(lp.bpms):
                bpm.__dict__["_buffers"]["reading"] = readings[k].reshape(lead + (2,))
            for screen, record, image in zip(lp.screens, records, images):
                if lead and image is not None:
                    image = image.reshape(lead + tuple(image.shape[-2:]))
                screen._record_stretch(record, N, sp, image, lead=lead)
            return ParticleBeam(out.reshape(p.shape) if lead else out, e_out, particle_charges=q, survival_probabilities=w_out, s=s_out,
                                species=sp), i + lp.count
        if lp.bpms or lp.apertures or lead:
            # active BPMs / apertures in the stretch (chx_lattice_track_diag): the particle pass leaves the weighted sums of x and
            # y at every monitor (one more launch forms all readings) and thins the survival probabilities at every aperture —
            # three launches for the lattice instead of six per monitor and a stop per aperture
            w = None
            if lp.bpms or lp.apertures:
                w = incoming.survival_probabilities
                if w.dtype != p.dtype or w.device != p.device or (torch.is_grad_enabled() and w.requires_grad) \
                        or w.shape[-1] != N or w.dim() > len(lead) + 1:
                    return None
                Bw = B
                if w.numel() == N and B > 1:
                    w, Bw = w.reshape(N), 1             # one row of weights for all beams: read as it is, not spread over the rows
                elif tuple(w.shape) != lead + (N,):
                    try:
                        w = w.expand(*lead, N)
                    except RuntimeError:
                        return None
                if not w.is_contiguous():
                    w = w.contiguous()
            for ap in lp.apertures:
                ap._check_limits()           # (aperture.py:72-73; a host read once per value of the two limits)
            n_bpm = len(lp.bpms)
            readings = ws = None
            ws_bytes = 0
            if n_bpm:
                readings = torch.empty((n_bpm, B, 2), dtype=p.dtype, device=p.device)
                ws_bytes = _lib.lib().chx_lattice_diag_workspace_bytes(N, B, n_bpm)
                ws = _ops.workspace(ws_bytes, p.device)
            if lp.apertures:
                w_out = torch.empty((*lead, N), dtype=p.dtype, device=p.device)
            out = None if lead == lead_x else torch.empty((*lead, N, 7), dtype=p.dtype, device=p.device)
            out, e_out, s_out = _HOST.lattice_track(lp.capsule, x, N, e, s_in if on_device else None, sp.mass_eV_float,
                                                    sp.num_elementary_charges_float, lp.device.index, w,
                                                    w_out if lp.apertures else None, n_bpm, readings, ws, ws_bytes, B,
                                                    Bx, Bm, Bw if w is not None else B, flags, out,
                                                    torch.empty(lead, dtype=p.dtype, device=p.device) if (lp.e_out_rows and not energy_rows) else None)
            if lp.e_out_rows and not energy_rows:
                here = self._lead_at(e.shape, lp.e_acc, ())
                if here != lead:                 # (only some axes of a grid scan reach the cavities' voltages and phases)
                    from .cavity import _narrow_to

                    e_out = _narrow_to(e_out, here)
            for k, bpm in enumerate(lp.bpms):
                r = readings[k].reshape(*lead, 2)
                if lead != lead_x:
                    # the beam AT the monitor is spread over the settings in front of it only (and over the energies once a map was
                    # applied): equal rows beyond that, the reading has the shape the walk's has — (2,) in front of the scan,
                    # (8, 1, 2) behind the first axis of a grid scan
                    here = lead if lp.bpm_acc[k] == lead else self._lead_at(lead_x, lp.bpm_acc[k], e.shape if (energy_rows and lp.bpm_after[k]) else ())
                    if here != lead:
                        from .cavity import _narrow_to

                        r = _narrow_to(r, (*here, 2))
                bpm.__dict__["_buffers"]["reading"] = r
            if lp.apertures and lead != lead_x:
                here = self._lead_at(torch.broadcast_shapes(lead_x, w_lead), lp.ap_acc[-1], e.shape if (energy_rows and lp.ap_after[-1]) else ())
                if here != lead:
                    from .cavity import _narrow_to

                    w_out = _narrow_to(w_out, (*here, N))     # (the beam at the LAST aperture is not spread over the whole scan yet)
        else:
            out, e_out, s_out = _HOST.lattice_track(lp.capsule, x, x.shape[0], e, s_in if on_device else None, sp.mass_eV_float,
                                                    sp.num_elementary_charges_float, lp.device.index)
        if s_out is None:
            s_out = s_in
            for kind, item in lp.items[:lp.count]:
                if kind == "run" or item._is_cavity:           # (monitors and apertures have no length)
                    s_out = s_out + (self._run_length(item) if kind == "run" else item.length)
        return ParticleBeam(out, e_out, particle_charges=incoming.particle_charges,
                            survival_probabilities=w_out, s=s_out, species=sp), i + lp.count

    @staticmethod
    def _stretch_grad_run(lp, incoming: ParticleBeam):
        """(run, its differentiable plan) when the stretch `lp` is [run of scalar settings | one active Screen] and the only
        things that carry a graph are settings of the run (a beam, an energy, a species or a screen geometry with a graph: None)."""
        p, e, sp = incoming.particles, incoming.energy, incoming.species
        if p.requires_grad or e.requires_grad or sp.mass_eV.requires_grad or sp.num_elementary_charges.requires_grad:
            return None
        if lp.count != 2 or len(lp.screens) != 1 or lp.bpms or lp.apertures or lp.vshape is not None or lp.items[0][0] != "run" \
                or lp.items[1][1] is not lp.screens[0] or p.dim() != 2 or e.dim() != 0:
            return None
        b = lp.screens[0].__dict__["_buffers"]
        if b["misalignment"].requires_grad or b["pixel_size"].requires_grad:
            return None
        run = lp.items[0][1]
        fr = run.gfast
        if fr is None or fr.dtype != p.dtype or fr.device != p.device:
            fr = run.gfast = _FastRun(run, p.dtype, p.device, allow_grad=True)
        elif fr.epoch != Element._epoch:
            fr.refresh()
        if not fr.ok or fr.grad_meta is None or fr.E != lp.shape[1]:
            return None
        if _CHECK_PLANS:
            fr.verify()
        return run, fr

    def _lattice_stretch_parameter(self, plan, i: int, incoming: ParameterBeam):
        """The stretch of `_lattice_stretch` for a ParameterBeam (`chx_parameter_lattice_track`: the same preparation launch, then
        one wavefront per batch row walks the items): (outgoing beam, index behind the stretch) or None. The walk item by item
        costs ~80-160 us of host time per cavity / monitor cell for 49 numbers of beam state."""
        mu, cov = incoming.mu, incoming.cov
        if not mu.is_cuda or cov.dtype != mu.dtype or cov.device != mu.device:
            return None
        cache = self.__dict__.get("_lattice_cache")
        if cache is None or cache[0] is not plan:
            cache = self._lattice_cache_for(plan)
        e = incoming.energy
        with_screens = Segment._STRETCH_SCREENS and mu.dim() == 1 and cov.dim() == 2 and e.dim() == 0       # (one beam: active Screens are items of the stretch)
        key = (i, mu.dtype, mu.device, "moments", with_screens)
        entry = cache[1].get(key)
        if entry is None:
            j, special = i, 0
            while j < len(plan) and (plan[j][0] == "run" or plan[j][1]._is_cavity or plan[j][1]._is_bpm or plan[j][1]._is_aperture
                                     or (with_screens and plan[j][1]._is_screen)):
                special += plan[j][0] != "run"
                j += 1
            entry = cache[1][key] = False if (j - i < 2 or special == 0) else [j, None]
        if entry is False:
            return None
        s_in, sp = incoming.s, incoming.species
        if e.dtype != mu.dtype or e.device != mu.device or (e.dim() != 0 and not e.is_contiguous()):
            return None
        energy_rows = e.dim() != 0          # a scan of beam energies: row b of the maps is built for energy b
        lp = entry[1]
        if lp is None or lp.epoch != Element._epoch:
            if torch.cuda.is_current_stream_capturing():
                return None
            if lp is None:
                lp = entry[1] = _LatticePlan(plan[i:entry[0]], mu.dtype, mu.device, allow_vector=True, allow_screens=with_screens)
            else:
                lp.refresh()
        if not lp.ok or lp.apertures:                # (an aperture only warns for a ParameterBeam: the walk does that)
            return None
        if _CHECK_PLANS:
            lp.verify()
        if torch.is_grad_enabled() and (mu.requires_grad or cov.requires_grad or e.requires_grad or sp.mass_eV.requires_grad
                                        or sp.num_elementary_charges.requires_grad or _any_requires_grad(*lp.tensors)):
            return None
        if lp.screens:
            # one ParameterBeam through [run | cavity | monitor | active Screen]+: the C++ host step allocates the outgoing moments,
            # every screen's record and its image (the bivariate normal density, screen.py:255-291) and enqueues the launches
            q = incoming.total_charge
            on_device = s_in.dim() == 0 and s_in.dtype == mu.dtype and s_in.device == mu.device and not s_in.requires_grad
            if not on_device or not mu.is_contiguous() or not cov.is_contiguous() or q.dim() != 0 or q.dtype != mu.dtype \
                    or q.device != mu.device or e.dtype != mu.dtype or e.device != mu.device \
                    or (torch.is_grad_enabled() and q.requires_grad):
                return None
            _ops.check_current_device(lp.device)
            geoms = []
            for screen in lp.screens:
                nx, ny = screen._geometry("sample_counts")
                geoms.append((screen._geometry("gauss_geom"), screen.__dict__["_buffers"]["misalignment"], nx, ny))
            n_bpm = len(lp.bpms)
            readings = torch.empty((n_bpm, 1, 2), dtype=mu.dtype, device=mu.device) if n_bpm else None
            mu_out, cov_out, e_out, s_out, records, images = _TORCH_HOST.parameter_lattice_track_screens(
                lp.capsule_s, mu, cov, e, s_in, q, sp.mass_eV_float, sp.num_elementary_charges_float, lp.device.index, tuple(geoms),
                n_bpm, readings)
            for k, bpm in enumerate(lp.bpms):
                bpm.__dict__["_buffers"]["reading"] = readings[k].reshape(2)
            for screen, record, image in zip(lp.screens, records, images):
                screen._record_stretch(record, 0, sp, image, "moments")
            return ParameterBeam(mu_out, cov_out, e_out, total_charge=q, s=s_out, species=sp), i + lp.count
        try:
            lead = torch.broadcast_shapes(mu.shape[:-1], cov.shape[:-2], lp.vshape if lp.vshape is not None else (), e.shape)
        except RuntimeError:
            return None
        B = _ops.numel(lead)
        if (lp.vshape is not None and tuple(lp.vshape) != tuple(lead)) or (energy_rows and tuple(e.shape) != tuple(lead)):
            return None
        Bm = B if (lp.vshape is not None or energy_rows) else 1
        if Bm > 1 and not lp.small_runs and Bm > Segment._STRETCH_MAX_ROWS:
            return None
        if Bm > 1 and not lp.ensure_rows(Bm):
            return None
        if lp.expanded:
            Segment._refresh_expanded(lp.expanded)
        m2 = mu.reshape(-1, 7) if mu.is_contiguous() else mu.reshape(-1, 7).contiguous()
        c2 = cov.reshape(-1, 49) if cov.is_contiguous() else cov.reshape(-1, 49).contiguous()
        if m2.shape[0] not in (1, B) or c2.shape[0] not in (1, B):
            return None
        _ops.check_current_device(lp.device)
        on_device = s_in.dim() == 0 and s_in.dtype == mu.dtype and s_in.device == mu.device and not s_in.requires_grad
        mu_out = torch.empty((*lead, 7), dtype=mu.dtype, device=mu.device)
        cov_out = torch.empty((*lead, 7, 7), dtype=mu.dtype, device=mu.device)
        e_out = torch.empty(tuple(lead), dtype=e.dtype, device=e.device) if (lp.e_out_rows and not energy_rows) else torch.empty_like(e)
        s_out = torch.empty_like(s_in) if on_device else None
        n_bpm = len(lp.bpms)
        readings = torch.empty((n_bpm, B, 2), dtype=mu.dtype, device=mu.device) if n_bpm else None
        n_items, n_elems, n_ptrs = lp.shape
        _ops.check(_lib.lib().chx_parameter_lattice_track(
            lp.table.data_ptr(), n_items, n_elems, n_ptrs, e.data_ptr(), sp.mass_eV_float, sp.num_elementary_charges_float, lp.code,
            lp.state.data_ptr(), lp.state.numel() * 8, m2.data_ptr(), c2.data_ptr(), B, m2.shape[0], c2.shape[0], Bm,
            lp.small_runs | (2 if energy_rows else 0) | (4 if lp.e_out_rows else 0),
            mu_out.data_ptr(),
            cov_out.data_ptr(), e_out.data_ptr(), s_in.data_ptr() if on_device else None, s_out.data_ptr() if on_device else None,
            n_bpm, readings.data_ptr() if n_bpm else None, _ops.stream_ptr()), "chx_parameter_lattice_track")
        in_lead = tuple(torch.broadcast_shapes(mu.shape[:-1], cov.shape[:-2])) if (mu.dim() > 1 or cov.dim() > 2) else ()
        lead_t = tuple(lead)
        if lp.e_out_rows and not energy_rows:
            here = self._lead_at(e.shape, lp.e_acc, ())
            if here != tuple(lead):
                from .cavity import _narrow_to

                e_out = _narrow_to(e_out, here)
        for k, bpm in enumerate(lp.bpms):
            r = readings[k].reshape(*lead, 2)
            if in_lead == lead_t or lp.bpm_acc[k] == lead_t:
                bpm.__dict__["_buffers"]["reading"] = r         # (nothing to narrow: the usual case, no shape arithmetic per monitor)
                continue
            here = self._lead_at(in_lead, lp.bpm_acc[k], e.shape if (energy_rows and lp.bpm_after[k]) else ())
            if here != lead_t:
                # the beam at the monitor is spread over the settings in FRONT of it only: equal rows beyond that, the reading has
                # the shape the walk's has
                from .cavity import _narrow_to

                r = _narrow_to(r, (*here, 2))
            bpm.__dict__["_buffers"]["reading"] = r
        if s_out is None:
            s_out = s_in
            for kind, item in lp.items[:lp.count]:
                if kind == "run" or item._is_cavity:
                    s_out = s_out + (self._run_length(item) if kind == "run" else item.length)
        return ParameterBeam(mu_out, cov_out, e_out, total_charge=incoming.total_charge, s=s_out, species=sp), i + lp.count

    @staticmethod
    def _identity_run(run) -> bool:
        """A run of pass-through elements only (Markers, inactive BPMs / Screens between two non-linear elements): nothing to
        apply, zero length."""
        return all(e._chx_kind == _IDENTITY and e._static_skippable and not e._parameters for e in run.elements)

    def _second_order_run(self, plan, i: int, incoming: ParticleBeam):
        """plan[i] and what follows it as one `chx_second_order_chain_mixed` call — second-order elements and the merged runs of
        linear elements between them (a lattice whose drifts are linear and whose magnets second order): (outgoing beam, index
        behind the stretch), or None when fewer than two items or no second-order element qualify (one plain beam without a
        graph, (7, 7, 7) maps and scalar lengths of the beam's dtype without gradients, the stock `track`; runs with scalar
        settings). Same numbers as the walk item by item."""
        x, energy, s = incoming.particles, incoming.energy, incoming.s
        if x.dim() != 2 or not x.is_cuda or energy.dim() != 0 or s.dim() != 0 or s.dtype != x.dtype or s.device != x.device or (
                torch.is_grad_enabled() and (x.requires_grad or energy.requires_grad or s.requires_grad)):
            return None
        grad = torch.is_grad_enabled()
        species = incoming.species
        # the stretch as it was found last time stands while no attribute of any element was assigned (`Element._epoch`), the
        # energy is the same tensor at the same version and no setting was edited in place: O(1) + one version sweep instead
        # of ~2.5 us of cache look-ups per element
        cache = self.__dict__.get("_so_run_cache")
        if cache is None or cache[0] is not plan:
            cache = self.__dict__["_so_run_cache"] = (plan, {})
        ent = cache[1].get(i)
        if ent is not None and ent["epoch"] == Element._epoch and ent["energy"] is energy and ent["energy_version"] == energy._version \
                and ent["dtype"] == x.dtype and ent["device"] == x.device and ent["mass"] == species.mass_eV_float \
                and ent["nq"] == species.num_elementary_charges_float \
                and [t._version for t in ent["tensors"]] == ent["versions"] and not _ops.CAPTURING[0] \
                and not (grad and any(t.requires_grad for t in ent["tensors"])):
            out, s_out, _ = _ops.second_order_chain(ent["maps"], ent["lengths"], x, s, ent["ptrs"])
            return ParticleBeam(out, energy, particle_charges=incoming.particle_charges,
                                survival_probabilities=incoming.survival_probabilities, s=s_out, species=species), ent["end"]
        maps, lengths, linear, tensors = [], [], [], []
        j, last_so = i, None
        while j < len(plan):
            kind, e = plan[j]
            if kind == "run":
                # a merged run of linear elements: its composed map (chx_run_map, one launch now, none while the stretch stands)
                # in a tensor of its own — the plan's state is shared with every other way this run can be tracked
                if j + 1 < len(plan) and plan[j + 1][0] == "element" and plan[j + 1][1]._is_cavity:
                    break                                      # that run belongs to the cavity's stretch (chx_lattice_track)
                if self._identity_run(e):
                    j += 1                                     # Markers between two magnets: nothing to apply, s + 0
                    continue
                zero = self.__dict__.get("_zero_s")
                if zero is None or zero.dtype != x.dtype or zero.device != x.device:
                    zero = self.__dict__["_zero_s"] = torch.zeros((), dtype=x.dtype, device=x.device)
                got = self._run_map_fast(e, x, energy, species, zero)
                if got is None or got[1].dim() != 0 or got[1].dtype != x.dtype or got[1].device != x.device:
                    break
                maps.append(got[0].clone())
                lengths.append(got[1])                         # 0 + (l_0 + l_1 + ...): the run's length as the walk adds it
                linear.append(1)
                tensors += e.fast.tensors
            elif kind == "element":
                cls = type(e)
                if e._tracking_method != "second_order" or e._t_kind is None or cls.track is not Element.track \
                        or cls._track_second_order is not Element._track_second_order \
                        or cls._track_internal is not Element._track_internal:
                    break
                length = e.length
                if length.dim() != 0 or length.dtype != x.dtype or length.device != x.device or (grad and length.requires_grad):
                    break
                T = e.second_order_transfer_map(energy, species)
                if T.dim() != 3 or T.dtype != x.dtype or T.device != x.device or (grad and T.requires_grad):
                    break
                maps.append(T if T.is_contiguous() else T.contiguous())
                lengths.append(length)
                linear.append(0)
                tensors += e._feature_key()[1]
                last_so = j
            else:
                break
            j += 1
        if last_so is None or len(maps) < 2:
            return None
        out, s_out, ptrs = _ops.second_order_chain(maps, lengths, x, s, linear=linear)
        if not _ops.CAPTURING[0] and not any(t.requires_grad for t in tensors) and not energy.requires_grad:
            # (maps of tensors that carry gradients are rebuilt on every track, Element._cached_map: nothing to keep)
            cache[1][i] = {"epoch": Element._epoch, "energy": energy, "energy_version": energy._version, "dtype": x.dtype,
                           "device": x.device, "mass": species.mass_eV_float, "nq": species.num_elementary_charges_float,
                           "tensors": tensors, "versions": [t._version for t in tensors], "maps": maps, "lengths": lengths,
                           "ptrs": ptrs, "end": j}
        return ParticleBeam(out, energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=s_out, species=species), j

    @staticmethod
    def _stable_energy(ent: dict, energy: torch.Tensor, mass: float, e_out: torch.Tensor) -> torch.Tensor:
        """The reference energy behind a cached drift-kick-drift stretch as the SAME tensor object from track to track while the
        incoming energy is the same tensor at the same version (then the value is the same): what stands downstream and depends
        on the energy — run maps of a later stretch, second-order maps — recognises it by identity and keeps its own products.
        A tensor somebody edited in place since is not handed out again."""
        last = ent.get("e_out")
        if last is not None and ent["e_in"] is energy and ent["e_in_version"] == energy._version and ent["e_mass"] == mass \
                and last._version == ent["e_out_version"]:
            return last
        ent["e_in"], ent["e_in_version"], ent["e_mass"], ent["e_out"], ent["e_out_version"] = energy, energy._version, mass, e_out, \
            e_out._version
        return e_out

    def _dkd_run(self, plan, i: int, incoming: ParticleBeam):
        """plan[i] and the drift-kick-drift elements behind it as one `chx_dkd_chain_mixed` call: (outgoing beam, index behind
        the run), or None when fewer than two items qualify (one plain beam without a graph; scalar settings of the beam's
        dtype that carry no gradient; the stock `track`) — then `Element.track` takes each of them as before. Drifts,
        Quadrupoles and Dipoles of one arithmetic mode keep the particles in registers, and the merged runs of linear elements
        between them (a lattice whose drifts are linear and whose magnets drift-kick-drift) ride in the same pass."""
        x, energy, s = incoming.particles, incoming.energy, incoming.s
        if x.dim() != 2 or not x.is_cuda or energy.dim() != 0 or energy.dtype != x.dtype or energy.device != x.device \
                or s.dim() != 0 or s.dtype != x.dtype or s.device != x.device or (
                    torch.is_grad_enabled() and (x.requires_grad or energy.requires_grad or s.requires_grad)):
            return None
        grad = torch.is_grad_enabled()
        species = incoming.species
        # the run as it was found last time stands while no attribute of any element was assigned (`Element._epoch`) and no
        # setting was edited in place (one sweep over the version counters instead of ~3 us of look-ups per element); with
        # linear runs inside also: the same energy tensor at the same version, the same species (their maps depend on both)
        cache = self.__dict__.get("_dkd_run_cache")
        if cache is None or cache[0] is not plan:
            cache = self.__dict__["_dkd_run_cache"] = (plan, {})
        ent = cache[1].get(i)
        if ent is not None and ent["epoch"] == Element._epoch and ent["dtype"] == x.dtype and ent["device"] == x.device \
                and [t._version for t in ent["tensors"]] == ent["versions"] and not _ops.CAPTURING[0] \
                and [e.dkd_precision for e in ent["dkd"]] == ent["precisions"] \
                and not (grad and any(t.requires_grad for t in ent["tensors"])) \
                and (ent["energy"] is None or (ent["energy"] is energy and ent["energy_version"] == energy._version
                                               and ent["mass"] == species.mass_eV_float
                                               and ent["nq"] == species.num_elementary_charges_float)):
            out, e_out, s_out, _ = _ops.dkd_chain(ent["kinds"], ent["params"], None, None, None, x, energy, s, species.mass_eV_float,
                                                  species.num_elementary_charges_float, ent["arrays"])
            e_out = self._stable_energy(ent, energy, species.mass_eV_float, e_out)
            return ParticleBeam(out, e_out, particle_charges=incoming.particle_charges,
                                survival_probabilities=incoming.survival_probabilities, s=s_out, species=species), ent["end"]
        # what stands at plan[i:]: drift-kick-drift elements that qualify (with their argument set) and runs of linear elements
        seq, after = [], []                                    # after[k]: the plan index behind item k (and the Markers behind it)
        j = i
        while j < len(plan):
            kind, e = plan[j]
            if kind == "run":
                if j + 1 < len(plan) and plan[j + 1][0] == "element" and plan[j + 1][1]._is_cavity:
                    break                                      # that run belongs to the cavity's stretch (chx_lattice_track)
                if self._identity_run(e):
                    j += 1                                     # Markers between two magnets: nothing to apply, s + 0
                    if after:
                        after[-1] = j
                    continue
                seq.append((e, None))
            elif kind == "element":
                cls = type(e)
                if e._tracking_method != "drift_kick_drift" or e._dkd_kind is None or cls.track is not Element.track \
                        or cls._track_drift_kick_drift is not Element._track_drift_kick_drift \
                        or cls._track_internal is not Element._track_internal or e.dkd_precision not in _ops.DKD_PRECISION:
                    break
                p = e._dkd_params_stacked(x.dtype, x.device)
                if p is None:
                    break
                n, f = e._dkd_options()
                seq.append((e, (e._dkd_kind, p, n, f, _ops.DKD_PRECISION[e.dkd_precision], [t for t, _ in e._dkd_scalar_refs()])))
            else:
                break
            j += 1
            after.append(j)
        # the arithmetic class of every item: a linear run goes with any (None); Drifts, Quadrupoles and Dipoles with their
        # `dkd_precision` (float64 beams are evaluated in fp64 whatever it says: one class); anything else has none (-1)
        single = x.dtype == torch.float64
        classes = [None if a is None else ((0 if single else a[4]) if a[0] in _DKD_IN_REGISTERS and x.dtype in (torch.float32, torch.float64)
                                           else -1) for _, a in seq]
        first = next((k for k, c in enumerate(classes) if c is not None), None)
        if first is None:
            return None
        k = 0
        if classes[first] >= 0:
            while k < len(seq) and classes[k] in (None, classes[first]):
                k += 1
        while k >= 2:
            # a stretch whose particles stay in registers: seq[:k]
            kinds = [_ops.DKD_LINEAR if a is None else a[0] for _, a in seq[:k]]
            has_runs = _ops.DKD_LINEAR in kinds
            params, lengths, steps, fringes, storage, tensors = [], [], [], [], [], []
            energies = _ops.dkd_energy_chain(kinds, energy, species.mass_eV_float) if has_runs else None
            failed = None
            for r, (obj, a) in enumerate(seq[:k]):
                if a is None:
                    # the run's composed map for the energy in front of it (chx_run_map: one launch now, none while the stretch
                    # stands) in a tensor of its own — the plan's state is shared with every other way this run can be tracked
                    zero = self.__dict__.get("_zero_s")
                    if zero is None or zero.dtype != x.dtype or zero.device != x.device:
                        zero = self.__dict__["_zero_s"] = torch.zeros((), dtype=x.dtype, device=x.device)
                    got = self._run_map_fast(obj, x, energy if r == 0 else energies[r - 1], species, zero)
                    if got is None or got[1].dim() != 0 or got[1].dtype != x.dtype or got[1].device != x.device:
                        failed = r
                        break
                    params.append(got[0].clone())
                    lengths.append(got[1])                     # 0 + (l_0 + l_1 + ...): the run's length as the walk adds it
                    steps.append(1)
                    fringes.append(0)
                    storage.append(0)
                    tensors += obj.fast.tensors
                else:
                    params.append(a[1])
                    lengths.append(None)
                    steps.append(a[2])
                    fringes.append(a[3])
                    storage.append(a[4])
                    tensors += a[5]
            if failed is not None:
                k = failed                                     # the stretch ends in front of the run that does not qualify
                continue
            if not any(a is not None for _, a in seq[:k]):
                return None
            out, e_out, s_out, arrays = _ops.dkd_chain(kinds, params, steps, fringes, storage, x, energy, s, species.mass_eV_float,
                                                       species.num_elementary_charges_float, lengths=lengths if has_runs else None)
            if not _ops.CAPTURING[0]:
                cache[1][i] = {"epoch": Element._epoch, "dtype": x.dtype, "device": x.device, "tensors": tensors,
                               "versions": [t._version for t in tensors], "kinds": kinds, "params": params, "lengths": lengths,
                               "arrays": arrays, "end": after[k - 1], "energy": energy if has_runs else None,
                               "energy_version": energy._version, "mass": species.mass_eV_float,
                               "nq": species.num_elementary_charges_float,
                               # (`dkd_precision` may be set on the CLASS, which moves no epoch: compared on every hit)
                               "dkd": [obj for obj, a in seq[:k] if a is not None],
                               "precisions": [obj.dkd_precision for obj, a in seq[:k] if a is not None]}
            return ParticleBeam(out, e_out, particle_charges=incoming.particle_charges,
                                survival_probabilities=incoming.survival_probabilities, s=s_out, species=species), after[k - 1]
        # no such stretch starts here: consecutive drift-kick-drift elements (no runs) in one call, element passes, up to where
        # a stretch could start
        if seq[0][1] is None:
            return None
        k = 1
        while k < len(seq) and seq[k][1] is not None and not (
                classes[k] >= 0 and k + 1 < len(seq) and classes[k + 1] in (None, classes[k])):
            k += 1
        if k < 2:
            return None
        kinds, params = [a[0] for _, a in seq[:k]], [a[1] for _, a in seq[:k]]
        tensors = [t for _, a in seq[:k] for t in a[5]]
        out, e_out, s_out, arrays = _ops.dkd_chain(kinds, params, [a[2] for _, a in seq[:k]], [a[3] for _, a in seq[:k]],
                                                   [a[4] for _, a in seq[:k]], x, energy, s, species.mass_eV_float,
                                                   species.num_elementary_charges_float)
        if not _ops.CAPTURING[0]:
            cache[1][i] = {"epoch": Element._epoch, "dtype": x.dtype, "device": x.device, "tensors": tensors,
                           "versions": [t._version for t in tensors], "kinds": kinds, "params": params, "arrays": arrays,
                           "end": after[k - 1], "energy": None, "dkd": [obj for obj, _ in seq[:k]],
                           "precisions": [obj.dkd_precision for obj, _ in seq[:k]]}
        return ParticleBeam(out, e_out, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=s_out, species=species), after[k - 1]

    @staticmethod
    def _next_chain_kick(plan, i: int, kick, dtype):
        """Index of the SpaceChargeKick that follows plan[i] through at most one run of linear elements and can continue its
        chain (same grid, chainable settings), else None."""
        j = i + 1
        if j < len(plan) and plan[j][0] == "run":
            j += 1
        if j < len(plan) and plan[j][0] == "element" and isinstance(plan[j][1], SpaceChargeKick) \
                and plan[j][1].grid_shape == kick.grid_shape and plan[j][1]._chain_settings_ok(dtype):
            return j
        return None

    #: a chain pays off while the beam keeps its deposit-tile order from kick to kick. The device counts the particles each
    #: deposit finds outside their slot's tile; the header with the running mean comes back ASYNCHRONOUSLY after the first
    #: tracks of a plan (pinned copy + event, polled at the next track: no synchronisation), and a plan whose beam reshuffles
    #: (a phase advance of tens of degrees between kicks) goes back to the kick-by-kick path, where the deposit sorts from
    #: scratch: at 25 % misfiled the two cost the same, at 100 % the chain's deposit is 350 us against 72.
    _CHAIN_MAX_MISFILED_PERMILLE = 250
    _CHAIN_SAMPLES = 3

    def _chain_guard(self, plan) -> dict:
        guard = self.__dict__.get("_chain_guard_state")
        if guard is None or guard["plan"] is not plan:
            guard = self.__dict__["_chain_guard_state"] = {"plan": plan, "off": False, "samples": 0, "pending": None, "host": None}
        return guard

    def _chain_allowed(self, plan) -> bool:
        if _CHAIN_MODE != "auto":        # CHX_SC_CHAIN=on / off pins the path (reproducible last bits from run to run)
            return _CHAIN_MODE == "on"
        guard = self._chain_guard(plan)
        pending = guard["pending"]
        if pending is not None and not torch.cuda.is_current_stream_capturing() and pending[1].query():   # (no polling inside a recording)
            header = pending[0]
            if int(header[6]) > 0 and int(header[3]) > self._CHAIN_MAX_MISFILED_PERMILLE * int(header[6]):
                guard["off"] = True
            guard["pending"] = None
        return not guard["off"]

    def _chain_report(self, plan, state) -> None:
        """Behind the last kick of a chain: fetch the chain's header for `_chain_allowed` (the first tracks of a plan only)."""
        guard = self._chain_guard(plan)
        if guard["off"] or guard["pending"] is not None or guard["samples"] >= self._CHAIN_SAMPLES:
            return
        if torch.cuda.is_current_stream_capturing():        # an event recorded into a graph cannot be polled later
            return
        guard["samples"] += 1
        host = guard["host"]
        if host is None:                                   # one page-locked buffer per plan (its allocation is the slow part)
            host = guard["host"] = torch.empty(8, dtype=torch.int32, pin_memory=True)
        host.copy_(state[:32].view(torch.int32), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        guard["pending"] = (host, done)

    def _chain_starts(self, plan, i: int, incoming: ParticleBeam) -> bool:
        kick = plan[i][1]
        dtype = incoming.particles.dtype
        if not (self._chain_allowed(plan) and kick._chain_settings_ok(dtype) and kick._chain_beam_ok(incoming)
                and self._next_chain_kick(plan, i, kick, dtype) is not None):
            return False
        if i + 1 < len(plan) and plan[i + 1][0] == "run" and self._chain_run_plan(plan[i + 1][1], incoming) is None:
            return False     # the run behind the first kick cannot ride in its particle pass: no second link to profit from
        return _lib.lib().chx_sc_tile_state_bytes(incoming.particles.shape[0], _ops._bins3(kick.grid_shape), _ops.dtype_code(dtype)) > 0

    @staticmethod
    def _chain_run_plan(run: _Run, incoming: ParticleBeam):
        """(persistent device plan of `run`, was it (re)built in this call) when the run can be applied inside a chain kick's
        particle pass — scalar settings of the beam's dtype on its device, nothing trainable or requiring grad, at most 192
        elements — else None."""
        p = incoming.particles
        fr = run.fast
        fresh = False          # the plan's device buffers were (re)written on the main stream just now
        if fr is None or fr.dtype != p.dtype or fr.device != p.device:
            fr = run.fast = _FastRun(run, p.dtype, p.device)
            fresh = True
        elif fr.epoch != Element._epoch:
            fr.refresh()
            fresh = True
        e, sp = incoming.energy, incoming.species
        if not fr.ok or e.dim() != 0 or e.dtype != fr.dtype or e.device != fr.device:
            return None
        if torch.is_grad_enabled() and (e.requires_grad or sp.mass_eV.requires_grad or sp.num_elementary_charges.requires_grad
                                        or _any_requires_grad(*fr.tensors)):
            return None
        return fr, fresh

    def _chain_kick(self, kick, run, fused, incoming: ParticleBeam, state, first: bool, last: bool, index: int = 0):
        """One link of a chain: the kick and, when the run behind it has a persistent device plan (`fused` from
        `_chain_run_plan`), that run in the same particle pass. Returns (beam, number of plan items consumed)."""
        R_addr, s_out = None, None
        if fused is not None:
            fr, fresh = fused
            p = incoming.particles
            e, sp = incoming.energy, incoming.species
            addr = ctypes.c_void_p()
            s_in = incoming.s
            s_out = self._device_s(fr, s_in)
            # The run's map (and path length) is built on the kick's SIDE stream, in front of the Green-function chain the
            # kick puts there: it depends on nothing the chain computes (settings, the reference energy), and the main stream
            # joins the side stream before the gather pass that applies the map. Off the main stream's critical path: 5 us
            # per kick. The side stream first waits for what the main stream did before the chain (first link: settings edited in
            # place, the beam) or to the plan's buffers (a plan built or refreshed in this very call).
            side = kick._chain_side_stream(p.device)
            stream = _ops.stream_ptr()
            if side is not None:
                if first or fresh:
                    side.wait_stream(torch.cuda.current_stream(p.device))
                stream = side.cuda_stream
            _ops.check(_lib.lib().chx_run_map(fr.kinds, fr.ptrs, fr.E, e.data_ptr(), sp.mass_eV_float,
                                              sp.num_elementary_charges_float, fr.code, fr.state.data_ptr(), fr.state_bytes,
                                              ctypes.byref(addr), s_in.data_ptr() if s_out is not None else None,
                                              s_out.data_ptr() if s_out is not None else None, stream), "chx_run_map")
            R_addr = addr.value
            if s_out is None:
                s_out = self._run_s(run, s_in)
        out = kick._track_in_chain(incoming, state, first, last, R_addr, index)
        beam = ParticleBeam(out, incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=s_out if fused is not None else incoming.s,
                            species=incoming.species)
        return beam, (2 if fused is not None else 1)

    def _kick_then_run(self, kick, run: _Run, incoming: ParticleBeam):
        """One particle pass for a SpaceChargeKick and the run of linear elements behind it (`chx_run_map` refreshes the run's
        stored map on the device, `chx_sc_kick` applies it to the kicked particles in registers): bit-identical to tracking
        the two one after the other. None when either side does not qualify."""
        p = incoming.particles
        if p.dim() != 2 or not p.is_cuda:
            return None
        fr = run.fast
        if fr is None or fr.dtype != p.dtype or fr.device != p.device:
            fr = run.fast = _FastRun(run, p.dtype, p.device)
        elif fr.epoch != Element._epoch:
            fr.refresh()
        e = incoming.energy
        if not fr.ok or e.dim() != 0 or e.dtype != fr.dtype or e.device != fr.device:
            return None
        sp = incoming.species
        if torch.is_grad_enabled() and (e.requires_grad or sp.mass_eV.requires_grad or sp.num_elementary_charges.requires_grad
                                        or _any_requires_grad(*fr.tensors)):
            return None
        R_addr = ctypes.c_void_p()
        s_in = incoming.s
        s_out = self._device_s(fr, s_in)
        _ops.check(_lib.lib().chx_run_map(fr.kinds, fr.ptrs, fr.E, e.data_ptr(), sp.mass_eV_float, sp.num_elementary_charges_float,
                                          fr.code, fr.state.data_ptr(), fr.state_bytes, ctypes.byref(R_addr),
                                          s_in.data_ptr() if s_out is not None else None,
                                          s_out.data_ptr() if s_out is not None else None, _ops.stream_ptr()), "chx_run_map")
        out = kick._track_then_map(incoming, R_addr.value)
        if out is None:
            return None
        return ParticleBeam(out, incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities,
                            s=s_out if s_out is not None else self._run_s(run, s_in), species=incoming.species)

    @tracking_call
    def track_moments(self, incoming: ParticleBeam, exact: bool = True) -> ParameterBeam:
        """Track a `ParticleBeam` and return only the outgoing beam's moments as a `ParameterBeam` (mu, cov,
        energy, total_charge, s, species). Same moments as `self.track(incoming).as_parameter_beam()`, but the last run
        of linear elements is fused with the moment reduction (`chx_track_moments`): the tracked particles of
        that run are never written — for a scan of B lattice settings over one shared beam that is the
        (B, N, 7) array (11.5 GB at B = 4096, N = 1e5).

        `exact=False` transports the moments of the beam entering the last run algebraically instead, mu' = R mu,
        Sigma' = R Sigma R^T (one `chx_moments` pass of the SHARED beam + one B-wide 7x7 kernel, `chx_parameter_track`):
        no per-setting pass over the particles at all. A linear map transports first and second moments exactly; what
        differs from `exact=True` is only that the tracked particles the reference would reduce are rounded to the beam dtype
        coordinate by coordinate — relative differences of ~1e-7 of a sigma in fp32 (1e-4 of a sigma is the bound asserted
        in tests/test_gpu_parity.py for strongly focused rows, where sigma itself is a small difference), 1e-15 in fp64."""
        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        plan = self._plan()
        last_run = plan[-1][1] if plan and plan[-1][0] == "run" else None
        for kind, item in plan[:-1] if last_run is not None else plan:
            if kind == "run":
                tm = self._run_map(item, incoming.energy, incoming.species)
                incoming = ParticleBeam(_ops.apply_map(incoming.particles, tm), incoming.energy,
                                        particle_charges=incoming.particle_charges,
                                        survival_probabilities=incoming.survival_probabilities,
                                        s=self._run_s(item, incoming.s), species=incoming.species)
            else:
                incoming = item.track(incoming)
        if last_run is None:
            return incoming._as_parameter_beam_same_species()
        tm = self._run_map(last_run, incoming.energy, incoming.species)
        if not exact:
            entering = incoming._as_parameter_beam_same_species()
            mu, cov = _ops.parameter_track(entering.mu, entering.cov, tm)
            return ParameterBeam(mu, cov, incoming.energy, total_charge=incoming.total_charge,
                                 s=self._run_s(last_run, incoming.s), species=incoming.species)
        mom = _ops.track_moments(incoming.particles, incoming.survival_probabilities, tm)
        return ParameterBeam._from_moment_vector(mom, incoming.particles.dtype, incoming.energy,
                                                 total_charge=incoming.total_charge,
                                                 s=self._run_s(last_run, incoming.s), species=incoming.species)

    @tracking_call
    def track_screen_reading(self, incoming: ParticleBeam) -> torch.Tensor:
        """Track a `ParticleBeam` and return the image of the segment's FINAL element, an active cloud-in-cell `Screen`
        — same numbers as `self.track(incoming); screen.reading`, but the last run of linear elements is fused with the
        deposit (`chx_cic_deposit_mapped`): for every (setting, particle) only x' and y' are evaluated (rows 0 and 2 of
        R x, the apply kernel's fma chain) and deposited straight into that setting's image. The (B, N, 7) tracked array
        is never written and the Screen's read beam is not recorded (the screen's own `reading` / `get_read_beam()` keep
        what the last `track` left there). Falls back to `track` + `reading` when the lattice does not end in
        [run of linear elements, active cloud-in-cell Screen] (screen.py:187-239, 327-339)."""
        from .screen import Screen

        if not isinstance(incoming, ParticleBeam):
            raise TypeError(f"Parameter incoming is of invalid type {type(incoming)}")
        plan = self._plan()
        screen = plan[-1][1] if plan and plan[-1][0] == "element" else None
        fused = (isinstance(screen, Screen) and screen.is_active and screen.method == "cloud-in-cell" and len(plan) >= 2
                 and plan[-2][0] == "run")
        if fused:
            from .. import sharding

            # `chx_cic_deposit_mapped` has no backward and deposits this rank's particles only: whenever anything that
            # reaches the image carries a graph (beam, charges, survival, the screen's geometry, any lattice setting) or
            # the beam is particle-sharded, `track` + `reading` is taken — it propagates gradients through CicDeposit and
            # sums the image over the ranks
            if sharding.active_group() is not None:
                fused = False
            elif torch.is_grad_enabled():
                sp = incoming.species
                fused = not (_any_requires_grad_py(incoming.particles, incoming.particle_charges,
                                                   incoming.survival_probabilities, incoming.energy, sp.mass_eV,
                                                   sp.num_elementary_charges, screen.pixel_size, screen.misalignment)
                             or any(p.requires_grad for p in self.parameters())
                             or any(b.requires_grad for b in self.buffers()))
        if not fused:
            self.track(incoming)
            last = self.elements[-1]
            if not isinstance(last, Screen):
                raise ValueError("track_screen_reading needs a Screen as the last element of the segment")
            return last.reading
        for kind, item in plan[:-2]:
            if kind == "run":
                tm = self._run_map(item, incoming.energy, incoming.species)
                incoming = ParticleBeam(_ops.apply_map(incoming.particles, tm), incoming.energy,
                                        particle_charges=incoming.particle_charges,
                                        survival_probabilities=incoming.survival_probabilities,
                                        s=self._run_s(item, incoming.s), species=incoming.species)
            else:
                incoming = item._track_internal(incoming)
        tm = self._run_map(plan[-2][1], incoming.energy, incoming.species)
        w, h = screen.effective_resolution
        return _ops.cic_deposit_mapped(incoming.particles, tm, (0, 2), (w, h), screen.extent.reshape(2, 2),
                                       charge=incoming.particle_charges, survival=incoming.survival_probabilities,
                                       shift=screen.misalignment, abs_charge=True, transpose_2d=True)

    def track_elementwise(self, incoming: ParticleBeam, fused: bool = False) -> ParticleBeam:
        """Track element by element WITHOUT merging transfer maps (every element is a real pass over
        the particles, results identical to `for e in elements: beam = e.track(beam)`). Runs of linear
        elements are dispatched as one `chx_track_elementwise` (E passes over HBM) or, with
        `fused=True`, one `chx_track_fused` call (one pass, particle kept in registers)."""
        for kind, item in self._plan():
            if kind == "run":
                if any(isinstance(e, Segment) for e in item.elements):
                    for e in item.elements:
                        incoming = e.track_elementwise(incoming, fused) if isinstance(e, Segment) else e.track(incoming)
                    continue
                stack = self._run_stack(item, incoming.energy, incoming.species)
                out = _ops.track_elementwise(incoming.particles, stack, fused=fused)
                incoming = ParticleBeam(out, incoming.energy, particle_charges=incoming.particle_charges,
                                        survival_probabilities=incoming.survival_probabilities,
                                        s=self._run_s(item, incoming.s), species=incoming.species)
            else:
                incoming = item.track_elementwise(incoming, fused) if isinstance(item, Segment) else item.track(incoming)
        return incoming

    def beam_along_segment_generator(self, incoming, resolution=None):
        """Beams at the end of every element, or every `resolution` metres (segment.py:631-656)."""
        if resolution is not None:
            yield from self.__class__(elements=self.split(resolution),
                                      name=f"{self.name}_split").beam_along_segment_generator(incoming)
        else:
            yield incoming
            for element in self.elements:
                incoming = element.track(incoming)
                yield incoming

    #: trailing dims of non-scalar beam attributes (beam.py `UNVECTORIZED_NUM_ATTR_DIMS`)
    _ATTR_DIMS = {"particles": 2, "particle_charges": 1, "survival_probabilities": 1, "mu": 1, "cov": 2, "x": 1, "px": 1,
                  "y": 1, "py": 1, "tau": 1, "p": 1, "energies": 1, "momenta": 1}

    def get_beam_attrs_along_segment(self, attr_names, incoming, resolution=None):
        """Beam attributes (every moment of a beam comes out of one fused `chx_moments` call per position) along the
        segment, stacked on a new axis in front of the attribute's own dims (segment.py:658-700)."""
        names = attr_names if isinstance(attr_names, tuple) else (attr_names,)
        if resolution is not None:
            return self.__class__(elements=self.split(resolution),
                                  name=f"{self.name}_split").get_beam_attrs_along_segment(attr_names, incoming)
        fast = self._attrs_along_fused(names, incoming)
        if fast is not None:
            return fast if isinstance(attr_names, tuple) else fast[0]
        per_beam = [tuple(getattr(beam, n) for n in names)
                    for beam in self.beam_along_segment_generator(incoming, resolution=resolution)]
        results = tuple(torch.stack(torch.broadcast_tensors(*[vals[i] for vals in per_beam]),
                                    dim=-(self._ATTR_DIMS.get(n, 0) + 1)) for i, n in enumerate(names))
        return results if isinstance(attr_names, tuple) else results[0]

    #: beam attributes that are functions of the first and second moments (plus energy / s / charge): available from ONE
    #: fused pass over the particles for all positions at once
    _MOMENT_ATTRS = frozenset(
        [f"mu_{c}" for c in ("x", "px", "y", "py", "tau", "p")] + [f"sigma_{c}" for c in ("x", "px", "y", "py", "tau", "p")]
        + ["cov_xpx", "cov_ypy", "cov_taup", "cov_xp", "cov_pxp", "cov_yp", "cov_pyp", "emittance_x", "emittance_y",
           "normalized_emittance_x", "normalized_emittance_y", "projected_emittance_x", "projected_emittance_y", "beta_x",
           "beta_y", "alpha_x", "alpha_y", "gamma_x", "gamma_y", "dispersion_x", "dispersion_px", "dispersion_y",
           "dispersion_py", "mu", "cov", "energy", "s", "total_charge", "relativistic_gamma", "relativistic_beta", "p0c"])

    def _attrs_along_fused(self, names, incoming):
        """Moment attributes after every element of an all-linear lattice from ONE pass over the particles: the prefix
        products R_e = M_e ... M_0 (`chx_compose_prefix`) go through `chx_track_moments` as a batch of E + 1 maps on the
        shared beam, instead of E tracking passes and E x len(names) reductions (segment.py:658-700) — for a ParameterBeam through
        `chx_parameter_track` as a batch of E + 1 maps on its one moment vector (element by element the walk costs ~0.45 ms per
        element: 44 ms for 100 elements). Positions are the ends of the segment's OWN elements: a nested segment counts once, its
        leaves' maps are part of the product. Returns None when the request does not qualify (other attributes, a vectorised or
        differentiable beam, non-linear or active elements)."""
        is_particles = isinstance(incoming, ParticleBeam)
        if not (is_particles or isinstance(incoming, ParameterBeam)) or not all(n in self._MOMENT_ATTRS for n in names):
            return None
        if not all(hasattr(ParameterBeam, n) or n in ("mu", "cov", "energy", "s", "total_charge") for n in names):
            return None
        if incoming.energy.dim() != 0:
            return None
        if is_particles:
            p = incoming.particles
            if p.dim() != 2 or incoming.survival_probabilities.dim() != 1 or incoming.particle_charges.dim() != 1 or p.requires_grad \
                    or not p.is_cuda:
                return None
        else:
            p = incoming.mu
            if p.dim() != 1 or incoming.cov.dim() != 2 or p.requires_grad or incoming.cov.requires_grad or not p.is_cuda:
                return None
        elements = list(self.elements)
        plan = self._plan()
        if not elements or not plan:
            return None
        bpms = []
        if len(plan) == 1 and plan[0][0] == "run":
            run = plan[0][1]
        else:
            # active BPMs between the runs let the beam pass (bpm.py:77-87): their place in the product is an identity, their
            # readings come from the moments at their position
            from .marker import BPM

            if not all(k == "run" or (item._is_bpm and type(item)._track_internal is BPM._track_internal and not item._parameters)
                       for k, item in plan):
                return None
            cached = self.__dict__.get("_along_cache")
            if cached is None or cached[0] is not plan:
                leaves = [e for k, item in plan for e in (item.elements if k == "run" else [item])]
                cached = self.__dict__["_along_cache"] = (plan, _Run(leaves))
            run = cached[1]
            bpms = [(k, e) for k, e in enumerate(run.elements) if e._is_bpm and e.is_active]
            if bpms:
                from .. import sharding

                if sharding.active_group() is not None:
                    return None       # (a particle-sharded beam: the monitors read GLOBAL means, BPM._track_internal exchanges them)
        # where the segment's own elements end in the run's list of leaves
        ends, n_leaves = [], 0
        for e in elements:
            n_leaves += len(e._leaves()) if type(e) is Segment and type(e).track is Segment.track else 1
            ends.append(n_leaves)
        if n_leaves != len(run.elements):
            return None
        lengths = [e.length for e in elements]
        if any(t is None or t.dim() != 0 or t.requires_grad for t in lengths):
            return None
        stack = self._run_stack(run, incoming.energy, incoming.species)          # (leaves, Bm, 7, 7)
        if stack.shape[1] != 1 or stack.requires_grad:
            return None
        eye = torch.eye(7, dtype=stack.dtype, device=stack.device).reshape(1, 1, 7, 7)
        prefix = torch.cat([eye, _ops.compose_prefix(stack)], dim=0).reshape(-1, 7, 7)        # position 0 = the incoming beam
        full = prefix
        if n_leaves != len(elements):
            prefix = prefix[torch.tensor([0] + ends, device=prefix.device)]
        s_along = incoming.s + torch.cat([torch.zeros_like(lengths[0]).reshape(1), torch.stack(lengths).cumsum(0)])
        if is_particles:
            mom = _ops.track_moments(p, incoming.survival_probabilities, prefix)                # (E + 1, 29)
            along = ParameterBeam._from_moment_vector(mom, p.dtype, incoming.energy, total_charge=incoming.total_charge, s=s_along,
                                                      species=incoming.species)
        else:
            if prefix.dtype != p.dtype:
                return None
            mu, cov = _ops.parameter_track(incoming.mu, incoming.cov, prefix)                  # (E + 1, 7), (E + 1, 7, 7)
            along = ParameterBeam(mu, cov, incoming.energy, total_charge=incoming.total_charge, s=s_along, species=incoming.species)
        if bpms:
            # the monitors' readings, as the walk would leave them: (mu_x, mu_y) of the beam at the monitor minus its misalignment
            at = torch.tensor([k + 1 for k, _ in bpms], device=full.device)
            if is_particles:
                xy = (mom if full is prefix else _ops.track_moments(p, incoming.survival_probabilities, full[at]))
                xy = (xy[at] if full is prefix else xy)[:, 2:5:2].to(p.dtype)
            else:
                m_at = mu[at] if full is prefix else _ops.parameter_track(incoming.mu, incoming.cov, full[at])[0]
                xy = m_at[:, 0:3:2]
            for row, (_, bpm) in zip(xy, bpms):
                bpm.__dict__["_buffers"]["reading"] = row - bpm.misalignment
        n_pos = prefix.shape[0]
        out = []
        for n in names:
            v = getattr(along, n)
            if v.dim() == 0 or v.shape[0] != n_pos:      # energy, total_charge: the same at every position
                v = v.expand(n_pos, *v.shape) if v.dim() else v.expand(n_pos)
            out.append(v)
        return tuple(out)

    def set_attrs_on_every_element(self, filter_type=None, is_recursive: bool = True, **kwargs) -> None:
        """segment.py:702-724"""
        for element in self.elements:
            if filter_type is None or isinstance(element, filter_type):
                for key, value in kwargs.items():
                    setattr(element, key, value)
            elif is_recursive and isinstance(element, Segment):
                element.set_attrs_on_every_element(filter_type=filter_type, is_recursive=True, **kwargs)

    # ---- lattice utilities (segment.py:73-367, 584-629) -----------------------------------------------------
    @property
    def element_names(self) -> list[str]:
        return [element.name for element in self.elements]

    def element_index(self, element_name: str) -> int:
        try:
            return self.element_names.index(element_name)
        except ValueError:
            raise ValueError(f"Element '{element_name}' not found in segment.")

    def subcell(self, start=None, end=None, include_start: bool = True, include_end: bool = True) -> "Segment":
        """Elements from `start` to `end` (names; None = the segment's own ends), segment.py:94-141."""
        names = self.__dict__["_by_name"]
        if start is not None and start not in names:
            raise ValueError(f"Element {start} is not part of the segment.")
        if end is not None and end not in names:
            raise ValueError(f"Element {end} is not part of the segment.")
        # by position: the cut ends at the first element called `end` (one that is also called `start` opens the cut instead, as
        # in the reference's walk), begins at the first element called `start` in front of it — or not at all when `start` only
        # comes later — and leaves out elements called `start` when include_start is off
        labels = [element.name for element in self.elements]
        stop = len(labels)
        if end is not None:
            stop = next((i for i, label in enumerate(labels) if label == end and label != start), stop)
        first = 0 if start is None else next((i for i in range(stop) if labels[i] == start), None)
        if first is None:
            return self.__class__([])
        picked = [element for element in self.elements[first:stop] if include_start or element.name != start]
        if include_end and stop < len(labels):
            picked.append(self.elements[stop])
        return self.__class__(picked)

    def flattened(self) -> "Segment":
        flat = []
        for e in self.elements:   # anything that can flatten itself does (sub-segments, Superimposed): segment.py:143-157
            flat += list(e.flattened().elements) if hasattr(e, "flattened") else [e]
        return self.__class__(elements=flat, name=self.name, sanitize_name=False)

    def reversed(self) -> "Segment":
        elements = [e.reversed() if isinstance(e, Segment) else e for e in self.elements][::-1]
        return self.__class__(elements=elements, name=f"{self.name}_reversed")

    def without_inactive_markers(self, except_for=None) -> "Segment":
        """All Markers dropped (the reference has no `is_active` for them either, segment.py:231-257)."""
        from .marker import Marker

        except_for = except_for or []
        return self.__class__(elements=[e for e in self.elements
                                        if not type(e) is Marker or e.name in except_for], name=self.name)

    def without_inactive_zero_length_elements(self, except_for=None) -> "Segment":
        except_for = except_for or []
        return self.__class__(
            elements=[e for e in self.elements
                      if bool((e.length != 0.0).any()) or getattr(e, "is_active", False) or e.name in except_for],
            name=self.name)

    def inactive_elements_as_drifts(self, except_for=None) -> "Segment":
        """Inactive elements with a length become Drifts of the same name (segment.py:288-324)."""
        from .drift import Drift

        except_for = except_for or []
        return self.__class__(
            elements=[e if getattr(e, "is_active", False) or bool((e.length == 0.0).all()) or e.name in except_for
                      else Drift(e.length, name=e.name, device=e.length.device, dtype=e.length.dtype)
                      for e in self.elements],
            name=self.name)

    def with_consecutive_elements_merged(self, except_for=None) -> "Segment":
        """Consecutive mergeable elements of one type combined (segment.py:326-367)."""
        except_for = except_for or []
        merged_elements = []
        current = self.elements[0]
        for next_element in list(self.elements)[1:]:
            if current.name not in except_for:
                if type(current) is Segment:
                    current = current.with_consecutive_elements_merged(except_for=except_for)
                elif type(current) is type(next_element) and next_element.name not in except_for:
                    merged = current.merge(next_element)
                    if merged is not None:
                        current = merged
                        continue
            merged_elements.append(current)
            current = next_element
        merged_elements.append(current)
        return self.__class__(elements=merged_elements, name=self.name, metadata=deepcopy(self.metadata))

    def _no_plot(self, *args, **kwargs):
        raise NotImplementedError("Segment.plot_*: plotting is outside this tracking engine (SURVEY.md section 2); "
                                  "`get_beam_attrs_along_segment` gives the numbers the reference's plots draw")

    plot_beam_attrs = plot_beam_attrs_over_lattice = plot_mean_and_std = plot_overview = plot_twiss = _no_plot
    plot_twiss_over_lattice = _no_plot

    def split(self, resolution) -> list[Element]:
        return [part for element in self.elements for part in element.split(resolution)]

    def merge(self, other: "Segment") -> "Segment":
        from .element import merge_element_names

        return self.__class__(elements=list(self.elements) + list(other.elements),
                              name=merge_element_names(self.name, other.name),
                              metadata={**other.metadata, **self.metadata})

    def partition_at(self, element_name: str, mode: str = "both"):
        """(pre, element, post) / (pre, post) around a named element (segment.py:599-629)."""
        index = self.element_index(element_name)
        elements = list(self.elements)
        pre_cell = self.__class__(elements[: index + 1]) if mode == "after" else self.__class__(elements[:index])
        post_cell = self.__class__(elements[index:]) if mode == "before" else self.__class__(elements[index + 1:])
        return (pre_cell, elements[index], post_cell) if mode == "both" else (pre_cell, post_cell)

    def transfer_maps_merged(self, incoming_beam: ParticleBeam, except_for=None) -> "Segment":
        """Merge runs of skippable elements into CustomTransferMaps (segment.py:179-229)."""
        from .custom_transfer_map import CustomTransferMap

        except_for = except_for or []
        merged, run = [], []
        beam = incoming_beam

        def flush(last=False):
            nonlocal run, beam
            # a single skippable element between two others stays what it is; at the END of the lattice the reference wraps
            # even a single one into a CustomTransferMap (segment.py:221-226) — kept
            if len(run) > 1 or (last and run):
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
        flush(last=True)
        return Segment(merged, name=self.name)

    @classmethod
    def from_ocelot(cls, cell, name=None, sanitize_names=None, device=None, dtype=None, **kwargs) -> "Segment":
        """Translate an Ocelot cell (a list of Ocelot elements) element by element (segment.py:404-445)."""
        from ..converters import ocelot

        converted = [ocelot.convert_element(el, sanitize_name=sanitize_names, device=device, dtype=dtype) for el in cell]
        return cls(converted, name=name, sanitize_name=sanitize_names, **kwargs)

    @classmethod
    def from_bmad(cls, bmad_lattice_file_path: str, environment_variables=None, sanitize_names=None, device=None,
                  dtype=None) -> "Segment":
        """Read a Bmad lattice file; the line named by its `use` statement is built (segment.py:447-479)."""
        from pathlib import Path

        from ..converters import bmad

        return bmad.convert_lattice(Path(bmad_lattice_file_path), environment_variables, sanitize_names, device, dtype)

    @classmethod
    def from_elegant(cls, elegant_lattice_file_path: str, name: str, sanitize_names=None, device=None,
                     dtype=None) -> "Segment":
        """Read the beam line `name` of an Elegant lattice file (segment.py:481-507)."""
        from pathlib import Path

        from ..converters import elegant

        return elegant.convert_lattice(Path(elegant_lattice_file_path), name, sanitize_names, device, dtype)

    @classmethod
    def from_nx_tables(cls, filepath) -> "Segment":
        """Read the NX-tables export of the ARES lattice (segment.py:509-523)."""
        from pathlib import Path

        from ..converters import nxtables

        return nxtables.convert_lattice(Path(filepath))

    @classmethod
    def from_lattice_json(cls, filepath: str, device=None, dtype=None) -> "Segment":
        """Load a LatticeJSON file (segment.py:369-384)."""
        from ..latticejson import load_cheetah_model

        return load_cheetah_model(filepath, device=device, dtype=dtype)

    def to_lattice_json(self, filepath: str, title: str | None = None,
                        info: str = "This is a placeholder lattice description") -> None:
        """Save as LatticeJSON (segment.py:386-402)."""
        from ..latticejson import save_cheetah_model

        save_cheetah_model(self, filepath, title, info)

    def clone(self) -> "Segment":
        return self.__class__(elements=[e.clone() for e in self.elements], name=self.name,
                              metadata=deepcopy(self.metadata))

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["elements"]

    def __repr__(self) -> str:
        # segment.py:1061-1082: the element list as torch prints a ModuleList; beyond five elements the first and last two
        n = len(self.elements)
        if n <= 5:
            listed = repr(self.elements)
        else:
            rows = [f"({i}): {self.elements[i]!r}" for i in (0, 1, n - 2, n - 1)]
            rows.insert(2, " \u22ee")
            listed = "ModuleList(\n  {0}\n)".format("\n  ".join(rows))
        return f"{self.__class__.__name__}(elements={listed}, name={self.name!r})"

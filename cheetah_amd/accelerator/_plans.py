"""The persistent plans behind `Segment.track` (moved out of segment.py in round 6; no behaviour of their own beyond what is said here).

  _Run          a maximal run of consecutive skippable elements (cheetah/accelerator/segment.py:545-574 merges such a run into one map)
                with its caches: tensor lists per element revision, the composed map against a token of version counters, the length;
  _FastRun      the run's persistent DEVICE plan for `chx_run_track` / `chx_run_map`: packed kinds and the addresses of the scalar
                settings (host arrays handed over by value), re-validated against `Element._epoch`, patched per element revision —
                and, for plain-tensor assignments that change nothing but an address, on the spot (`absorb`, Element._absorb);
  _LatticePlan  the table of a STRETCH [run | active Cavity | BPM | Aperture | Screen]+ for the `chx_lattice_track*` calls: item rows,
                element kinds, pointer offsets and the settings' addresses in one int64 array on the device, with the host image and
                the patch list that lets a control step re-send it without re-deriving it.

What a cached map or plan may assume is the reference's cache contract (cheetah/utils/cache.py:6-68): a setting changes by assignment
(the epoch moves) or in place on its tensor (the device reads the value through its address)."""

from __future__ import annotations

import ctypes
import weakref

import torch

from .. import _lib, _ops
from .element import Element

_IDENTITY = _ops.KIND["identity"]


class _FastRun:
    """Persistent device plan of a run (`chx_run_track`): packed kinds / parameter pointers (host arrays, forwarded by
    value) and the device state buffer that remembers the settings the stored map was built from. Valid while
    `Element._epoch` stands still; a changed VALUE of a setting (in-place edit, optimiser step) is found by the device.

    A control loop re-assigns a few settings per step, which moves the epoch: `refresh` then re-reads only the elements
    whose own revision moved, patches their pointers into the existing host arrays and keeps the device state (it
    remembers VALUES, whatever tensor they live in).

    Contract: a setting changes by ASSIGNMENT (`quad.k1 = t`, moves the epoch) or by an in-place op on its tensor (same
    storage, the device sees the new value). Swapping the storage under an unchanged tensor object without an assignment
    (`buf.data = other`, `set_`, `resize_`) moves no counter the host looks at; the plan then still holds the old address.
    `CHX_CHECK_PLANS=1` re-derives every address on every track and raises on a mismatch (debug aid, ~25 us per track of a
    100-element run). The device state (`state`, `R_view`) is per run, not per stream: one lattice object is tracked from
    one stream at a time, like the reference's non-reentrant element caches (utils/cache.py:23-27)."""

    __slots__ = ("epoch", "ok", "dtype", "device", "kinds", "ptrs", "E", "state", "state_bytes", "tensors", "code",
                 "elements", "revs", "rows", "per_tensors", "slots", "R_view", "allow_grad", "distinct", "grad_slots", "capsule", "grad_meta",
                 "layout", "hooked", "__weakref__")

    def __init__(self, run, dtype, device, allow_grad=False):
        # allow_grad: the plan of the DIFFERENTIABLE run map (_ops.RunMapPlanned) — trainable parameters and settings that
        # require grad qualify; `distinct` / `grad_slots` then say which slots every distinct setting tensor feeds
        self.allow_grad = allow_grad
        self.distinct, self.grad_slots = (), ()
        self.capsule = None
        self.grad_meta = None
        self.layout = 0           # moves whenever the packed arrays are rebuilt (other kinds / another number of elements)
        self.hooked = False       # the elements know this plan (Element._hooks): re-assigned settings are patched in on the spot
        self.dtype, self.device = dtype, device
        self.elements = [e for e in run.elements]
        self.revs = [None] * len(self.elements)
        self.rows = [None] * len(self.elements)          # per element: (kind, [pointers]) or "identity"
        self.per_tensors = [()] * len(self.elements)
        self.state = None
        self.kinds = None
        self.slots = None
        self.R_view = None
        self.code = _ops.dtype_code(dtype)
        self.refresh()

    def _read(self, e, i):
        """(kind, pointers, tensors) of element number i, "identity", or None when the element rules the plan out. A setting
        that still is the tensor OBJECT read last time keeps its slot unexamined (same dtype, device, shape and address): a
        control step that re-assigns one strength of a quadrupole re-checks one tensor, not five."""
        if not e._plannable():
            return None                     # data-dependent skippability (an active Cavity, a sub-Segment)
        # (trainable parameters and settings that require grad are read like any other: whether a track may USE this plan is
        # decided per call — every user asks `torch.is_grad_enabled() and _any_requires_grad(*plan.tensors)` — so a model with
        # nn.Parameter strengths evaluated under no_grad keeps its plans: 75 -> 22 us for a 100-element lattice)
        kind = e._chx_kind
        if kind is None:
            return None
        if kind == _IDENTITY:
            return "identity"
        refs = e._builder_scalar_refs()
        old_row, old_tensors = self.rows[i], self.per_tensors[i]
        reuse = old_row is not None and old_row != "identity" and old_row[0] == kind and len(old_tensors) == len(refs)
        row = list(old_row[1]) if reuse else [None] * _ops.MAX_PARAMS
        tensors = []
        dtype, device = self.dtype, self.device
        for k, (t, index) in enumerate(refs):
            tensors.append(t)
            if reuse and t is old_tensors[k]:
                # the same tensor OBJECT: its storage may still have been swapped (`t.data = ...`, `set_`, `resize_`)
                if row[k] == (t.data_ptr() if index is None else t.data_ptr() + index * t.element_size()):
                    continue
            if t.dtype != dtype or t.device != device:
                return None
            if index is None:
                if t.dim() != 0:
                    return None
                row[k] = t.data_ptr()
            else:
                if t.dim() != 1 or not t.is_contiguous():
                    return None
                row[k] = t.data_ptr() + index * t.element_size()
        return kind, row, tensors

    def refresh(self) -> None:
        self.epoch = Element._epoch
        self.ok = False
        revs, rows = self.revs, self.rows
        changed, same_layout = [], self.kinds is not None
        for i, e in enumerate(self.elements):
            rev = e.__dict__["_revision"]
            if rev != revs[i]:
                got = self._read(e, i)
                if got is None:
                    rows[i] = None
                    revs[i] = None     # look again next time
                    self.kinds = None  # whatever is patched later starts from a full rebuild
                    return
                old = rows[i]
                if got == "identity":
                    same_layout = same_layout and old == "identity"
                    rows[i], self.per_tensors[i] = got, ()
                else:
                    same_layout = same_layout and old is not None and old != "identity" and old[0] == got[0]
                    rows[i], self.per_tensors[i] = (got[0], got[1]), tuple(got[2])
                revs[i] = rev
                changed.append(i)
            elif rows[i] is None:
                return
        if same_layout:
            # same kinds in the same places (the usual control step): patch the pointers of the elements that changed
            ptrs, slots = self.ptrs, self.slots
            for i in changed:
                r = rows[i]
                if r != "identity":
                    base = slots[i] * _ops.MAX_PARAMS
                    for k, v in enumerate(r[1]):
                        ptrs[base + k] = v
        else:
            kinds, pointers, slots = [], [], [None] * len(rows)
            for i, r in enumerate(rows):
                if r != "identity":
                    slots[i] = len(kinds)
                    kinds.append(r[0])
                    pointers += r[1]
            E = len(kinds)
            if E == 0 or E > 192:
                self.kinds = None
                return
            self.E, self.slots = E, slots
            self.layout += 1
            self.kinds = (ctypes.c_int32 * E)(*kinds)
            self.ptrs = (ctypes.c_void_p * (E * _ops.MAX_PARAMS))(*pointers)
            if not self.allow_grad:
                self.state_bytes = _lib.lib().chx_run_state_bytes(E)
                self.state = torch.full((self.state_bytes // 8,), float("nan"), dtype=torch.float64, device=self.device)
                # the plan as the C host step sees it (cheetah_amd._chxhost): addresses of the two arrays above and of the state
                self.capsule = _lib.host().plan(ctypes.addressof(self.kinds), ctypes.addressof(self.ptrs), E,
                                                self.state.data_ptr(), self.state_bytes, self.code) if self.state.is_cuda else None
            self.R_view = None
        # kept alive: the plan holds their addresses (a tensor may appear more than once: `misalignment` feeds two parameters)
        tensors = tuple([t for ts in self.per_tensors for t in ts])
        if len(tensors) > 400:
            return
        self.tensors = tensors
        if self.allow_grad:
            where, distinct, grad_slots = {}, [], []
            for i, e in enumerate(self.elements):
                if rows[i] == "identity":
                    continue
                for k, (t, index) in enumerate(e._builder_scalar_refs()):
                    pos = where.get(id(t))
                    if pos is None:
                        pos = where[id(t)] = len(distinct)
                        distinct.append(t)
                        grad_slots.append([])
                    grad_slots[pos].append((self.slots[i], k, index))
            self.distinct, self.grad_slots = tuple(distinct), tuple(tuple(g) for g in grad_slots)
            # the same plan as the C++ node reads it (cheetah_amd._chxtorch RunScreenTrack): one list of integers
            meta = [self.E, self.code, len(distinct)] + [int(k) for k in self.kinds] + [int(v or 0) for v in self.ptrs]
            for g in grad_slots:
                meta.append(len(g))
                for e, k, index in g:
                    meta += [e, k, -1 if index is None else index]
            self.grad_meta = meta
        self.ok = True
        if not self.allow_grad and not self.hooked:
            # the elements hand re-assigned setting tensors to this plan on the spot (Element._absorb -> absorb below)
            me = weakref.ref(self)
            for i, e in enumerate(self.elements):
                if rows[i] != "identity":
                    d = e.__dict__
                    hooks = d.get("_hooks")
                    if hooks is None:
                        hooks = d["_hooks"] = []
                    elif len(hooks) > 8:
                        hooks[:] = [h for h in hooks if h[0]() is not None]     # (plans of lattices that are gone)
                    hooks.append((me, i))
            self.hooked = True

    def absorb(self, e, i: int, refs, slots, rev: int, before: int) -> None:
        """Element number `i` (`e`) had the setting tensor behind its parameter slots `slots` replaced by one of the same dtype,
        device and shape (Element.__setattr__, a "soft" assignment at epoch `before` + 1): patch the addresses in place. A plan
        that was valid before the assignment stays valid; any other one re-reads its elements at the next track as before."""
        row = self.rows[i]
        if not self.ok or self.elements[i] is not e or self.revs[i] != rev - 1 or row is None or row == "identity" \
                or len(refs) != len(self.per_tensors[i]):
            return          # (a plan that has not seen this element's previous revision re-reads it at its next refresh)
        base = self.slots[i] * _ops.MAX_PARAMS
        ptrs, addresses = self.ptrs, row[1]
        tensors = list(self.per_tensors[i])
        for k, (t, index) in enumerate(refs):
            a = t.data_ptr() if index is None else t.data_ptr() + index * t.element_size()
            if k in slots:
                tensors[k] = t
            elif t is not tensors[k]:
                return      # (another setting is a different tensor object by now: not this assignment's doing — re-read the element)
            # the element's OTHER settings are looked at as well, like a re-read would: a storage swapped under an unchanged
            # tensor object (`t.data = ...`, outside the contract) is picked up by the next assignment on its element, as before
            if addresses[k] != a:
                ptrs[base + k] = a
                addresses[k] = a
        self.per_tensors[i] = tuple(tensors)
        self.revs[i] = rev
        if self.epoch == before:
            self.epoch = before + 1      # valid before the assignment: valid after it
        # (else: something else moved the epoch since this plan's last refresh — another lattice being built, say; the next
        # refresh compares revisions, finds this element up to date and re-reads only what it has not seen)
        # (users ask `_any_requires_grad(*plan.tensors)` per track: the list must name the tensors the plan addresses NOW)
        self.tensors = tuple([t for ts in self.per_tensors for t in ts])

    def verify(self) -> None:
        """CHX_CHECK_PLANS=1: every stored address against the tensor it was read from."""
        k = 0
        for i, e in enumerate(self.elements):
            r = self.rows[i]
            if r is None or r == "identity":
                continue
            for j, (t, index) in enumerate(e._builder_scalar_refs()):
                want = t.data_ptr() if index is None else t.data_ptr() + index * t.element_size()
                if self.ptrs[self.slots[i] * _ops.MAX_PARAMS + j] != want:
                    raise RuntimeError(f"run plan of element {e.name!r}: the storage of setting {j} was replaced without an "
                                       "attribute assignment (`.data = ...`, `set_`, `resize_`); assign the tensor instead")
                k += 1


class _LatticePlan:
    """Persistent device plan of a STRETCH of lattice — [run of skippable elements | active Cavity]+ with scalar settings — for
    `chx_lattice_track`: the whole stretch is two launches (every map, coefficient row, energy and the path length in one, every
    particle through all items in registers in the other) and one C call of the host step, where the element-by-element walk
    (segment.py:545-574 of the reference) costs two launches and ~20-30 us of host time per item. The table holds device
    ADDRESSES of the settings (read by the device on every track: in-place edits are followed); like `_FastRun` it is valid while
    `Element._epoch` stands still and is re-derived after any attribute assignment."""

    __slots__ = ("items", "count", "dtype", "device", "epoch", "ok", "table", "state", "capsule", "tensors", "code", "bpms", "apertures", "shape", "vshape", "allow_vector", "small_runs", "bpm_vec", "ap_vec", "bpm_after", "ap_after", "e_out_rows", "expanded", "bpm_acc", "ap_acc", "e_acc", "screens", "allow_screens", "capsule_s", "words", "patch", "others", "other_tensors", "words_np", "edge_keys")

    def __init__(self, items, dtype, device, allow_vector=False, allow_screens=False):
        self.items, self.dtype, self.device = items, dtype, device
        self.allow_vector = allow_vector       # a ParameterBeam's stretch takes runs with vectorised settings; particles end there
        self.allow_screens = allow_screens     # ONE plain beam under scalar settings: active Screens are items of the stretch
        self.screens = ()                      # the active screens of the stretch, in record-slot order
        self.edge_keys = ()                    # histogram screens: (screen, pixel-size tensor, its version) the edge arrays were formed from
        self.capsule_s = None
        self.code = _ops.dtype_code(dtype)
        self.table = self.state = self.capsule = None
        self.words = None
        self.ok = False
        self.patch = self.others = self.words_np = None
        self.other_tensors = ()
        self.tensors = ()
        self.bpms = ()          # the active BPMs of the stretch, in reading-slot order
        self.apertures = ()     # its active apertures
        self.vshape = None      # batch shape of vectorised settings inside the stretch (a ParameterBeam's stretch takes them)
        self.count = 0          # leading items the table covers (the stretch ends in front of the first item it cannot take)
        self.refresh()

    def _refresh_patched(self) -> bool:
        """The usual control step — a few settings re-ASSIGNED (`quad.k1 = tensor`: the epoch moves, the layout of the lattice does
        not): the runs' persistent plans patch the addresses of the elements that changed, those addresses are copied into the host
        image of the table at the places recorded when it was built, and the table is uploaded again — instead of re-deriving every
        row, kind and offset of the stretch (≈ 55 us of Python for the 13-element README section; this: ≈ 15 us)."""
        for item, rev in self.others:
            if item.__dict__["_revision"] != rev:
                return False                       # (a cavity / monitor / aperture / screen was touched: its own rows may differ)
        kept = []
        for fr, layout, view, src, dst in self.patch:
            if fr.epoch != Element._epoch:
                fr.refresh()
            if not fr.ok or fr.layout != layout:
                return False
            self.words_np[dst] = view[src]
            kept += fr.tensors
        # (a table of up to 448 words rides in the arguments of a one-workgroup launch, chx_table_store: no page-locked staging
        # tensor, no copy call — ~12 us of a control step's host time)
        if not _lib.host().table_store(self.words_np, self.table.data_ptr(), self.device.index or 0):
            staging = torch.empty(self.words_np.shape[0], dtype=torch.int64, pin_memory=True)
            staging.numpy()[:] = self.words_np
            self.table.copy_(staging, non_blocking=True)
        self.tensors = tuple(kept) + self.other_tensors
        self.words = None                          # (verify() compares against a fresh derivation: the list form is rebuilt there)
        self.epoch = Element._epoch
        return True

    def refresh(self) -> None:
        from .cavity import Cavity
        from .segment import Segment

        if self.ok and self.patch is not None and not torch.cuda.is_current_stream_capturing() and self._refresh_patched():
            return
        self.epoch = Element._epoch
        self.ok = False
        self.patch = None
        lib = _lib.lib()
        dtype, device = self.dtype, self.device
        patch, others, other_tensors, patchable = [], [], [], True
        rows, elem_kind, elem_poff, ptrs, tensors, bpms, apertures = [], [], [], [], [], [], []
        screens, screen_shapes, edge_keys = [], [], []
        count = cavities = longest_run = 0
        vshape, bpm_vec, ap_vec = None, [], []
        bpm_after, ap_after, maps_seen = [], [], False   # does a run / cavity (a map) sit in front of the monitor / aperture?
        e_out_rows = False       # a cavity with a vectorised voltage or phase: the outgoing energy has the batch shape
        # the ONE batch shape of the stretch: what every vectorised setting in it broadcasts to ((8, 1) and (1, 8) of a grid scan:
        # (8, 8)); `acc`: what the settings in front of an item broadcast to — the batch shape the beam has there in the walk
        common, acc, expanded, bpm_acc, ap_acc = None, None, [], [], []
        e_acc = None             # what the vectorised voltages and phases broadcast to: the batch shape of the outgoing energy
        if self.allow_vector:
            found = []
            for kind, item in self.items:
                if kind == "run":
                    for e in item.elements:
                        if getattr(e, "_chx_kind", None) is None:
                            continue
                        found += [tuple(t.shape) for t, index in e._builder_scalar_refs() if index is None and t.dim() != 0]
                elif item._is_cavity:
                    found += [tuple(t.shape) for t in item._settings("voltage", "phase", "frequency") if t.dim() != 0]
            if found:
                try:
                    common = tuple(torch.broadcast_shapes(*found))
                except RuntimeError:
                    common = None           # (shapes that do not broadcast: the walk raises like the reference)

        def grown(a, b):
            return b if a is None else a if b is None else tuple(torch.broadcast_shapes(a, b))

        for kind, item in self.items:
            if kind != "run" and item._is_screen:
                # an active screen: {4, flags, where the addresses of its misalignment and pixel size (and its resolution and
                # image shape) sit in ptrs, record slot} — it records the beam that reaches it (and deposits its cloud-in-cell
                # image from the particle pass) and lets the beam pass (screen.py:187-239)
                from .screen import Screen

                b = item.__dict__["_buffers"]
                mis, ps = b.get("misalignment"), b.get("pixel_size")
                if not self.allow_screens or vshape is not None or len(screens) >= 4 \
                        or type(item)._track_internal is not Screen._track_internal or type(item).reading is not Screen.reading \
                        or not item.is_active or item.is_blocking or mis is None or ps is None \
                        or any(t.shape != (2,) or t.dtype != dtype or t.device != device or not t.is_contiguous() for t in (mis, ps)):
                    break
                res, bins = item.resolution, item.effective_resolution
                # the particle pass deposits the image: 1 cloud-in-cell (the extent derived on the device from the pixel size), 2
                # histogram (torch.histogramdd's bins on the edges torch.linspace gives: the edge arrays are formed here, on the
                # host's side of the call, and their addresses ride behind the screen's other entries)
                deposit = 1 if item.method == "cloud-in-cell" else 2 if item.method == "histogram" else 0
                rows += [4, deposit, len(ptrs), len(screens)]
                ptrs += [mis.data_ptr(), ps.data_ptr(), int(res[0]), int(res[1]), int(bins[0]), int(bins[1])]
                tensors += [mis, ps]
                others.append((item, item.__dict__["_revision"]))
                other_tensors += [mis, ps]
                if deposit == 2:
                    ex, ey = item.pixel_bin_edges
                    if ex.dtype != dtype or ey.dtype != dtype or not ex.is_contiguous() or not ey.is_contiguous():
                        break
                    ptrs += [ex.data_ptr(), ey.data_ptr()]
                    tensors += [ex, ey]
                    other_tensors += [ex, ey]
                    edge_keys.append((item, ps, ps._version))          # (an in-place edit of the pixel size: other edges)
                screens.append(item)
                screen_shapes.append((deposit, int(bins[0]), int(bins[1])))
                count += 1
                continue
            if kind != "run" and item._is_aperture:
                # an active aperture: {3, shape, where the addresses of x_max and y_max sit in ptrs, -}
                from .marker import Aperture

                limits = (item.x_max, item.y_max)
                if type(item)._track_internal is not Aperture._track_internal or not item.is_active \
                        or item.shape not in ("rectangular", "elliptical") \
                        or any(t.dim() != 0 or t.dtype != dtype or t.device != device for t in limits):
                    break
                rows += [3, 1 if item.shape == "elliptical" else 0, len(ptrs), 0]
                ptrs += [t.data_ptr() for t in limits]
                tensors += limits
                others.append((item, item.__dict__["_revision"]))
                other_tensors += limits
                apertures.append(item)
                ap_vec.append(vshape is not None)
                ap_after.append(maps_seen)
                ap_acc.append(acc)
                count += 1
                continue
            if kind != "run" and item._is_bpm:
                # an active beam position monitor: {2, 0, where its misalignment's address sits in ptrs, reading slot}
                from .marker import BPM

                mis = item.misalignment
                if type(item)._track_internal is not BPM._track_internal or not item.is_active \
                        or mis.shape != (2,) or mis.dtype != dtype or mis.device != device or not mis.is_contiguous():
                    break
                rows += [2, 0, len(ptrs), len(bpms)]
                ptrs.append(mis.data_ptr())
                tensors.append(mis)
                others.append((item, item.__dict__["_revision"]))
                other_tensors.append(mis)
                bpms.append(item)
                bpm_vec.append(vshape is not None)     # does a run with vectorised settings sit in front of this monitor?
                bpm_after.append(maps_seen)
                bpm_acc.append(acc)
                count += 1
                continue
            if kind == "run":
                if Segment._identity_run(item):
                    count += 1          # Markers / inactive diagnostics between two monitors: nothing to apply, no length
                    continue
                fr = item.fast
                if fr is None or fr.dtype != dtype or fr.device != device:
                    fr = item.fast = _FastRun(item, dtype, device)
                elif fr.epoch != Element._epoch:
                    fr.refresh()
                run_patch = None
                if fr.ok:
                    run_kinds = [fr.kinds[e] for e in range(fr.E)]
                    row_ptrs, src = [], []
                    for e in range(fr.E):
                        base = e * _ops.MAX_PARAMS
                        n_par = lib.chx_kind_num_params(fr.kinds[e])
                        row_ptrs.append([fr.ptrs[base + j] for j in range(n_par)])
                        src += range(base, base + n_par)
                    run_tensors = fr.tensors
                    run_patch = (fr, src, len(ptrs))          # (this run's addresses start at ptrs[len(ptrs)], in `src` order)
                else:
                    patchable = False
                    # settings vectorised over a batch of lattice settings: addresses tagged with their lowest bit (a (rows,) array);
                    # one batch shape for the whole stretch
                    got = Segment._vector_run_rows(item, dtype, device, common) if (self.allow_vector and not screens) else None
                    # (one workgroup per item and row prepares the maps: beyond a few hundred rows the walk item by item is cheaper)
                    if got is None or got[4] is None or (vshape is not None and got[4] != vshape) or len(got[0]) > 192 \
                            or _ops.numel(got[4]) > 65535:
                        break
                    run_kinds, vrows, vflags, run_tensors, vshape, run_expanded, own = got
                    expanded += run_expanded
                    acc = grown(acc, own)
                    if any(len(r) != lib.chx_kind_num_params(k) for r, k in zip(vrows, run_kinds)):
                        break
                    row_ptrs = [[q | f for q, f in zip(r, fl)] for r, fl in zip(vrows, vflags)]
                if not run_kinds or any(q is None for r in row_ptrs for q in r):
                    break
                longest_run = max(longest_run, len(run_kinds))
                maps_seen = True
                rows += [0, len(run_kinds), len(elem_kind), 0]
                for e, r in enumerate(row_ptrs):
                    elem_kind.append(run_kinds[e])
                    elem_poff.append(len(ptrs))
                    ptrs += r
                tensors += run_tensors
                if run_patch is not None:
                    patch.append(run_patch)
            else:
                if type(item).track is not Cavity.track:
                    break
                settings = item._settings("length", "voltage", "phase", "frequency")
                if any(t.dtype != dtype or t.device != device for t in settings):
                    break
                # a PHASE (voltage, frequency) scan: the setting is a tensor of the stretch's one batch shape, its address tagged
                # like a vectorised magnet strength; the cavity then hands on one energy per row
                cav_ptrs, cav_shape, fits, cav_expanded = [], vshape, True, []
                for k, t in enumerate(settings):
                    if t.dim() == 0:
                        cav_ptrs.append(t.data_ptr())
                    elif not self.allow_vector or screens or k == 0 or not t.is_contiguous() or common is None or _ops.numel(common) > 65535:
                        fits = False
                        break
                    else:
                        cav_shape = common
                        if tuple(t.shape) != common:           # (a phase of shape (8, 1) in a grid scan: an expanded copy)
                            with torch.no_grad():
                                copy = t.expand(common).contiguous()
                            cav_expanded.append((t, copy, [t._version]))
                            tensors.append(copy)
                            cav_ptrs.append(copy.data_ptr() | 1)
                        else:
                            cav_ptrs.append(t.data_ptr() | 1)
                        acc = grown(acc, tuple(t.shape))
                        if k in (1, 2):
                            e_out_rows, e_acc = True, grown(e_acc, tuple(t.shape))
                if not fits:
                    break
                vshape = cav_shape
                expanded += cav_expanded
                rows += [1, 1, len(elem_kind), 0]
                elem_kind.append(_ops.KIND[item._kind_name()])
                elem_poff.append(len(ptrs))
                ptrs += cav_ptrs
                tensors += settings
                others.append((item, item.__dict__["_revision"]))
                other_tensors += settings
                cavities += 1
                maps_seen = True
            count += 1
        # (a trailing run stays in the stretch: it rides in the same particle pass; a trailing BPM reads the outgoing beam)
        self.count = count
        self.bpms, self.apertures, self.bpm_vec, self.ap_vec = tuple(bpms), tuple(apertures), tuple(bpm_vec), tuple(ap_vec)
        self.bpm_after, self.ap_after = tuple(bpm_after), tuple(ap_after)
        self.e_out_rows = e_out_rows
        self.expanded, self.bpm_acc, self.ap_acc, self.e_acc = tuple(expanded), tuple(bpm_acc), tuple(ap_acc), e_acc
        self.screens, self.capsule_s = tuple(screens), None
        self.edge_keys = tuple(edge_keys)
        if count < 2 or (cavities == 0 and not bpms and not apertures and not screens) or not elem_kind:
            return
        n_items, n_elems, n_ptrs = len(rows) // 4, len(elem_kind), len(ptrs)      # (identity runs hold no row)
        self.vshape = vshape
        # a wave per (item, row) prepares a stretch without cavities and with short runs; else a workgroup does, which pays up to a
        # few hundred rows of vectorised settings only
        self.small_runs = 1 if (cavities == 0 and longest_run <= 64) else 0
        if longest_run <= 64:
            self.small_runs |= 8      # CHX_LATTICE_SHORT_RUNS: a wave per (item, row) also with cavities
        if vshape is not None and not self.small_runs and _ops.numel(vshape) > Segment._STRETCH_MAX_ROWS:
            return
        state_bytes = lib.chx_lattice_state_bytes_batched(n_items, n_elems, _ops.numel(vshape) if vshape is not None else 1)
        if state_bytes == 0:
            return
        # host -> device without a synchronisation: page-locked staging buffer (torch's caching host allocator keeps it alive
        # until the copy has run), asynchronous copy on the current stream
        words = rows + elem_kind + elem_poff + ptrs
        self.words = words
        if patchable and not expanded and count == len(self.items):
            import numpy as np

            base = len(rows) + len(elem_kind) + len(elem_poff)
            self.words_np = np.asarray(words, dtype=np.int64)
            self.patch = [(fr, fr.layout, np.frombuffer(fr.ptrs, dtype=np.int64), np.asarray(src, dtype=np.int64),
                           base + at + np.arange(len(src), dtype=np.int64)) for fr, src, at in patch]
            self.others, self.other_tensors = others, tuple(other_tensors)
        staging = torch.empty(len(words), dtype=torch.int64, pin_memory=True)
        staging.copy_(torch.tensor(words, dtype=torch.int64))
        if self.table is None or self.table.numel() != len(words):
            self.table = torch.empty(len(words), dtype=torch.int64, device=device)
        self.table.copy_(staging, non_blocking=True)
        if self.state is None or self.state.numel() * 8 < state_bytes:
            self.state = torch.empty(state_bytes // 8 + 1, dtype=torch.float64, device=device)
        self.capsule = _lib.host().lattice_plan(self.table.data_ptr(), n_items, n_elems, n_ptrs, self.state.data_ptr(),
                                                self.state.numel() * 8, self.code)
        if screens:
            # the same plan as the C++ host step sees it (cheetah_amd._chxtorch): with the screens' image shapes
            self.capsule_s = _lib.torch_host().stretch_plan(self.table.data_ptr(), n_items, n_elems, n_ptrs, self.state.data_ptr(),
                                                            self.state.numel() * 8, self.code, tuple(screen_shapes))
        self.shape = (n_items, n_elems, n_ptrs)
        self.tensors = tuple(tensors)       # kept alive: the table holds their addresses
        self.ok = True


    def verify(self) -> None:
        """CHX_CHECK_PLANS=1: the table re-derived from the elements as they are now must be the table on the device (a storage
        swapped under an unchanged tensor object — `.data = ...`, `set_`, `resize_` — moves no counter the host looks at)."""
        if self.expanded or torch.cuda.is_current_stream_capturing():
            return                  # (expanded copies of broadcast settings live in the plan itself: a fresh plan has other addresses)
        for kind, item in self.items[:self.count]:
            if kind == "run" and item.fast is not None and item.fast.ok:
                item.fast.verify()  # (the table takes a run's addresses from its persistent plan)
        fresh = _LatticePlan(self.items, self.dtype, self.device, allow_vector=self.allow_vector, allow_screens=self.allow_screens)
        mine = self.words if self.words is not None else [int(v) for v in self.words_np]
        if fresh.ok != self.ok or (fresh.ok and fresh.words != mine):
            raise RuntimeError("lattice stretch plan: the storage of a setting was replaced without an attribute assignment "
                               "(`.data = ...`, `set_`, `resize_`); assign the tensor instead")

    def ensure_rows(self, rows: int) -> bool:
        """Room for `rows` rows of maps in the device state (a scan of beam energies brings its rows with the BEAM, not with the
        lattice): the state grows when needed and the C host step's capsule follows it."""
        n_items, n_elems, n_ptrs = self.shape
        need = _lib.lib().chx_lattice_state_bytes_batched(n_items, n_elems, rows)
        if need == 0:
            return False
        if self.state.numel() * 8 < need:
            if torch.cuda.is_current_stream_capturing():
                return False          # (persistent state is not allocated from a recording's private pool: the walk is capturable)
            self.state = torch.empty(need // 8 + 1, dtype=torch.float64, device=self.device)
            self.capsule = _lib.host().lattice_plan(self.table.data_ptr(), n_items, n_elems, n_ptrs, self.state.data_ptr(),
                                                    self.state.numel() * 8, self.code)
        return True




class _Run:
    """A maximal run of consecutive skippable elements plus its caches. The run object survives changes of element
    SETTINGS (only a change of the element list or of skippability re-plans the segment), so what does not depend on the
    settings that changed — the tensor lists of the untouched elements, the summed length — is not redone."""

    __slots__ = ("elements", "modules", "rev", "per_module", "tensors", "params", "token", "tm", "stack", "length",
                 "length_key", "energy_ref", "s_cache", "fast", "gfast", "parts", "vrows")

    def __init__(self, elements):
        self.elements = elements
        self.parts = None         # a run too long for ONE persistent device plan: its pieces (Segment._run_map_parts)
        self.vrows = None         # the packed tables of Segment._run_map_vector, valid while the epoch stands still
        self.modules = [m for e in elements for m in e.modules() if isinstance(m, Element)]
        self.rev = None
        self.per_module = [None] * len(self.modules)   # (revision, buffers + parameters, parameters) per module
        self.tensors = []  # every buffer / parameter tensor of the run (for _version scans)
        self.params = []
        self.token = None
        self.tm = None
        self.stack = None
        self.length = None
        self.length_key = None
        self.energy_ref = None
        self.s_cache = None
        self.fast = None
        self.gfast = None
        self._collect()

    def _collect(self):
        """Refresh the tensor lists of the modules whose revision moved (an attribute was assigned)."""
        rev = tuple([m.__dict__["_revision"] for m in self.modules])
        if rev == self.rev:
            return
        self.rev = rev
        tensors, params = [], []
        for i, m in enumerate(self.modules):
            cached = self.per_module[i]
            if cached is None or cached[0] != rev[i]:
                own_params = [p for p in m._parameters.values() if p is not None]
                cached = (rev[i], [t for t in m._buffers.values() if t is not None] + own_params, own_params)
                self.per_module[i] = cached
            tensors += cached[1]
            params += cached[2]
        self.tensors, self.params = tensors, params

    def current_token(self, energy, species):
        self._collect()
        return (id(energy), energy._version, species.mass_eV_float, species.num_elementary_charges_float, self.rev,
                tuple([t._version for t in self.tensors]), tuple([t.requires_grad for t in self.tensors]))

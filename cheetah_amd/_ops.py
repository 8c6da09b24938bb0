"""Tensor-level wrappers of the libchx C-ABI (device pointers + current HIP stream) and the
``torch.autograd.Function``s whose backward passes are HIP kernels as well.

PyTorch is plumbing here: it owns the device memory (caching allocator) and the stream. All
arithmetic on particle-sized or map-sized data happens inside libchx.
"""

from __future__ import annotations

import ctypes
import weakref
import os

import torch

from . import _lib
from ._lib import CicArgs, Hist2dArgs, check

KIND = {
    "identity": 0, "drift": 1, "quadrupole": 2, "dipole": 3, "hcor": 4, "vcor": 5, "ccor": 6,
    "cavity_sw": 7, "cavity_tw": 8, "solenoid": 9, "undulator": 10,
}
NUM_PARAMS = [0, 1, 5, 9, 2, 2, 3, 4, 4, 4, 4]
MOM_NOUT = 29
CAV_NCOEF = 8


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return 0
    if dtype == torch.float64:
        return 1
    raise TypeError(f"cheetah_amd supports float32 and float64 beams, got {dtype}")


try:  # raw C accessors: ~0.3 us instead of ~4 us for torch.cuda.current_stream().cuda_stream (called per launch)
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _current_device = torch._C._cuda_getDevice
except AttributeError:  # pragma: no cover - torch build without the private accessors
    _raw_stream = None


def require_device(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "cheetah_amd tracks on the GPU only (HIP kernels, no CPU fallback): move the beam "
                "and the lattice to a ROCm device first, e.g. `.to('cuda')`."
            )
    check_current_device(tensors[0].device if tensors and tensors[0] is not None else None)


def check_current_device(device) -> None:
    """The kernels are launched on the CURRENT device's stream (one process per GPU is the design): tensors that live on
    another GPU of the same process must be tracked under `torch.cuda.device(...)`. Loud instead of a cross-device launch."""
    if device is not None and _raw_stream is not None and device.index is not None and device.index != _current_device():
        raise RuntimeError(f"the tensors live on {device} but the current device is cuda:{_current_device()}: "
                           f"run the call under `with torch.cuda.device({device.index}):`")


def stream_ptr() -> int:
    if _raw_stream is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def aligned(t: torch.Tensor) -> torch.Tensor:
    """Contiguous and 16-byte aligned view/copy of `t`."""
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def numel(shape) -> int:
    n = 1
    for s in shape:
        n *= int(s)
    return n


def bshapes(*shapes):
    """torch.broadcast_shapes with a fast path for the common case (all non-empty shapes equal); the
    torch implementation is a ~7 us Python function, which is as long as the apply kernel itself."""
    first = None
    for sh in shapes:
        if len(sh) == 0:
            continue
        if first is None:
            first = sh
        elif sh != first:
            return torch.broadcast_shapes(*shapes)  # also raises RuntimeError for improper shapes
    return torch.Size(()) if first is None else first


def flat_bcast(t: torch.Tensor, batch_shape, n_tail: int):
    """Flatten the leading (vector) dims of `t` against `batch_shape`.

    Returns (tensor of shape (Bt, *tail), Bt) with Bt in {1, B}; data is only materialised when the
    vector dims are a genuine partial broadcast.
    """
    tail = t.shape[t.dim() - n_tail:] if n_tail else ()
    lead = t.shape[: t.dim() - n_tail]
    B = numel(batch_shape)
    if numel(lead) == 1:
        return t.reshape(1, *tail), 1
    if tuple(lead) == tuple(batch_shape):
        return t.reshape(B, *tail), B
    return t.expand(*batch_shape, *tail).reshape(B, *tail), B


#: > 0 while a tracking step is being recorded into a device graph (`cheetah_amd.graph.capture`): the host-side caches that
#: skip a launch because "the values have not changed since" (maps keyed on version counters, stacked parameter arrays, memoised
#: moments) are bypassed — a replay re-runs the recorded launches against whatever the tensors hold THEN, so every kernel that
#: derives something from a setting must be part of the recording
CAPTURING = [0]
#: (with `capture(..., constant_beam=True)`: the moments of the beam entering the step may stay memoised)
CAPTURE_KEEPS_BEAM_MOMENTS = [False]


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class _CloneInto(torch.autograd.Function):
    """`t.clone()` whose copy is made by someone else (clone_many's one launch for all tensors): forward hands out the
    uninitialised destination, backward is the identity like CloneBackward."""

    @staticmethod
    def forward(ctx, t, dst):
        # the destination itself (declared modified in place), not a view of it: an in-place edit of the copy under autograd
        # must not trip over "a view was created inside a custom Function"
        ctx.mark_dirty(dst)
        return dst

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def clone_many(tensors):
    """Copies of up to 8 device tensors by ONE launch (chx_copy_arrays) instead of one `clone()` kernel each — while a device
    graph records also for tensors that carry an autograd graph (their copies are attached to it by an identity node); tensors
    that are not contiguous or do not live on a ROCm device are cloned the ordinary way."""
    # (a graph-carrying tensor joins the shared launch only while a device graph records — one kernel less in the replay; in
    # eager mode torch's own clone() is the cheaper call on the host, which is what bounds an eager step)
    graphs = torch.is_grad_enabled() and not CAPTURING[0]
    plain = [t.is_cuda and t.is_contiguous() and not (graphs and t.requires_grad) for t in tensors]
    out = [torch.empty_like(t) if ok else t.clone() for t, ok in zip(tensors, plain)]
    pairs = [(t, o) for t, o, ok in zip(tensors, out, plain) if ok and t.numel()]
    for lo in range(0, len(pairs), 8):
        chunk = pairs[lo:lo + 8]
        n = len(chunk)
        check(_lib.lib().chx_copy_arrays((ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in chunk]),
                                         (ctypes.c_void_p * n)(*[o.data_ptr() for _, o in chunk]),
                                         (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t, _ in chunk]), n, stream_ptr()),
              "chx_copy_arrays")
    if torch.is_grad_enabled():
        for i, (t, ok) in enumerate(zip(tensors, plain)):
            if ok and t.requires_grad:
                out[i] = _CloneInto.apply(t, out[i])
    for t, o in zip(tensors, out):
        # provenance (see `_LinearSource` / `_origin`): a copy of a linearly tracked array still IS R x; a copy of a weight
        # array still holds the values of its source
        lin = getattr(t, "_chx_lin", None)
        if lin is not None and lin.version == t._version:
            o._chx_lin = lin.rebound(o)
        # (the ROOT of a chain of copies, held weakly: a copy of a copy must not keep every array in between — or its source
        # — alive for as long as it lives itself)
        root = _origin(t)
        o._chx_origin = (weakref.ref(root), root._version, o._version)
    return out


# ---------------------------------------------------------------------------------------------
# map builders
def build_rmatrix_raw(kind: int, params, energy, mass_eV: float, n_charges: float, B: int) -> torch.Tensor:
    """params (Bp,P) / energy (Be,) device tensors of one dtype -> R (B,7,7)."""
    require_device(energy)
    dt = dtype_code(energy.dtype)
    R = torch.empty((B, 7, 7), dtype=energy.dtype, device=energy.device)
    Bp = params.shape[0] if params is not None else 1
    check(_lib.lib().chx_build_rmatrix(kind, ptr(params), ptr(energy), mass_eV, n_charges, B, Bp,
                                       energy.shape[0], dt, ptr(R), stream_ptr()), "chx_build_rmatrix")
    return R


class BuildMap(torch.autograd.Function):
    """R = builder(params, energy); backward = chx_build_rmatrix_vjp (dual numbers on device)."""

    @staticmethod
    def forward(ctx, params, energy, kind, mass_eV, n_charges, B):
        params = params.contiguous()
        energy = energy.contiguous()
        ctx.save_for_backward(params, energy)
        ctx.meta = (kind, mass_eV, n_charges, B)
        return build_rmatrix_raw(kind, params, energy, mass_eV, n_charges, B)

    @staticmethod
    def backward(ctx, dR):
        params, energy = ctx.saved_tensors
        kind, mass_eV, n_charges, B = ctx.meta
        P = NUM_PARAMS[kind]
        dR = dR.contiguous()
        dparams = torch.empty((B, max(P, 1)), dtype=energy.dtype, device=energy.device)
        denergy = torch.empty((B,), dtype=energy.dtype, device=energy.device)
        check(_lib.lib().chx_build_rmatrix_vjp(kind, ptr(params), ptr(energy), mass_eV, n_charges, ptr(dR),
                                               B, params.shape[0], energy.shape[0], dtype_code(energy.dtype),
                                               ptr(dparams), ptr(denergy), stream_ptr()), "chx_build_rmatrix_vjp")
        dparams = dparams[:, :P]
        if params.shape[0] == 1 and B > 1:
            dparams = dparams.sum(dim=0, keepdim=True)
        if energy.shape[0] == 1 and B > 1:
            denergy = denergy.sum(dim=0, keepdim=True)
        return dparams, denergy, None, None, None, None


def build_rmatrix(kind: int, params, energy, mass_eV, n_charges, B) -> torch.Tensor:
    if (params is not None and params.requires_grad) or energy.requires_grad:
        if params is None:
            params = energy.new_zeros((1, 0))
        return BuildMap.apply(params, energy, kind, mass_eV, n_charges, B)
    with torch.no_grad():
        return build_rmatrix_raw(kind, None if params is None else params.contiguous(), energy.contiguous(),
                                 mass_eV, n_charges, B)


MAX_PARAMS = 9  # CHX_MAX_PARAMS


def _scalar_run_tables(elements, energy: torch.Tensor):
    """(kinds, refs) of a run whose builder parameters are all device scalars of the energy's dtype, identity elements left
    out: refs[e] = [(tensor, index or None)] per parameter. None when the run does not qualify."""
    dtype, device = energy.dtype, energy.device
    kinds, refs_all = [], []
    identity = KIND["identity"]
    for e in elements:
        kind = e._chx_kind
        if kind is None:
            return None
        if kind == identity:
            continue
        refs = e._builder_scalar_refs()
        for t, index in refs:
            if t.dtype != dtype or t.device != device:
                return None
            if index is None:
                if t.dim() != 0:
                    return None
            elif t.dim() != 1 or not t.is_contiguous():
                return None
        kinds.append(kind)
        refs_all.append(refs)
    return kinds, refs_all


def _scalar_run_pointers(refs_all):
    pointers = []
    for refs in refs_all:
        row = [None] * MAX_PARAMS
        for k, (t, index) in enumerate(refs):
            row[k] = t.data_ptr() if index is None else t.data_ptr() + index * t.element_size()
        pointers += row
    return (ctypes.c_void_p * len(pointers))(*pointers)


def _build_compose_raw(kinds, pointers, energy, mass_eV, n_charges):
    """(element maps (E,7,7), composed map (7,7)) by chx_build_rmatrix_scalars + chx_compose_maps."""
    E = len(kinds)
    dtype, device = energy.dtype, energy.device
    lib = _lib.lib()
    code = dtype_code(dtype)
    maps = torch.empty((E, 7, 7), dtype=dtype, device=device)
    check(lib.chx_build_rmatrix_scalars((ctypes.c_int32 * E)(*kinds), pointers, E, ptr(energy), mass_eV, n_charges, code,
                                        ptr(maps), stream_ptr()), "chx_build_rmatrix_scalars")
    if E == 1:
        return maps, maps[0]
    base, step = maps.data_ptr(), 49 * maps.element_size()
    out = torch.empty((7, 7), dtype=dtype, device=device)
    check(lib.chx_compose_maps((ctypes.c_void_p * E)(*[base + e * step for e in range(E)]), (ctypes.c_uint8 * E)(*([1] * E)),
                               E, 1, code, ptr(out), stream_ptr()), "chx_compose_maps")
    return maps, out


class RunMapScalars(torch.autograd.Function):
    """Composed map of a run of scalar-parameter elements WITH gradients: forward = the two C calls of the no-grad path,
    backward = chx_run_vjp (compose backward + dual-number builders, two launches) — one autograd node for the whole run
    instead of a BuildMap node per element and a matmul node per product."""

    @staticmethod
    def forward(ctx, meta, energy, *tensors):
        kinds, slots, mass_eV, n_charges = meta           # slots[e][k] = (position in `tensors`, index or None)
        refs_all = [[(tensors[pos], index) for pos, index in row] for row in slots]
        pointers = _scalar_run_pointers(refs_all)
        maps, out = _build_compose_raw(kinds, pointers, energy, mass_eV, n_charges)
        ctx.meta = meta
        ctx.save_for_backward(energy, maps, *tensors)
        return out if len(kinds) > 1 else out.clone()

    @staticmethod
    def backward(ctx, dT):
        kinds, slots, mass_eV, n_charges = ctx.meta
        energy, maps, *tensors = ctx.saved_tensors
        E = len(kinds)
        refs_all = [[(tensors[pos], index) for pos, index in row] for row in slots]
        pointers = _scalar_run_pointers(refs_all)
        lib = _lib.lib()
        ws_bytes = lib.chx_run_vjp_workspace_bytes(E)
        ws = workspace(ws_bytes, energy.device)
        dT = dT.to(energy.dtype).contiguous()
        d = torch.empty((E, MAX_PARAMS + 1), dtype=energy.dtype, device=energy.device)
        check(lib.chx_run_vjp((ctypes.c_int32 * E)(*kinds), pointers, E, ptr(energy), mass_eV, n_charges, dtype_code(energy.dtype),
                              ptr(maps), ptr(dT), ptr(d), ptr(ws), ws_bytes, stream_ptr()), "chx_run_vjp")
        grads = [None] * len(tensors)
        for e, row in enumerate(slots):
            for k, (pos, index) in enumerate(row):
                if not ctx.needs_input_grad[2 + pos]:
                    continue
                t = tensors[pos]
                g = d[e, k]
                if index is not None:
                    full = grads[pos] if grads[pos] is not None else torch.zeros_like(t)
                    full[index] += g
                    grads[pos] = full
                else:
                    grads[pos] = g if grads[pos] is None else grads[pos] + g
        d_energy = d[:, MAX_PARAMS].sum() if ctx.needs_input_grad[1] else None
        return (None, d_energy, *grads)


class RunMapPlanned(torch.autograd.Function):
    """RunMapScalars on a persistent plan (`segment._FastRun(..., allow_grad=True)`): the packed kinds / pointer arrays and
    the (tensor -> slots) table are read once per change of the lattice, every step is ONE C call forward
    (chx_run_build_compose) and ONE backward (chx_run_vjp_masked: only the dual-number builders of the settings that want a
    gradient are evaluated)."""

    @staticmethod
    def forward(ctx, plan, energy, mass_eV, n_charges, *tensors):
        E = plan.E
        maps = torch.empty((E, 7, 7), dtype=energy.dtype, device=energy.device)
        out = torch.empty((7, 7), dtype=energy.dtype, device=energy.device)
        check(_lib.lib().chx_run_build_compose(plan.kinds, plan.ptrs, E, energy.data_ptr(), mass_eV, n_charges, plan.code,
                                               maps.data_ptr(), out.data_ptr(), stream_ptr()), "chx_run_build_compose")
        ctx.plan, ctx.consts = plan, (mass_eV, n_charges, plan.kinds, plan.ptrs, plan.grad_slots, plan.epoch)
        ctx.save_for_backward(energy, maps, *tensors)
        return out

    @staticmethod
    def backward(ctx, dT):
        mass_eV, n_charges, kinds, ptrs, grad_slots, epoch = ctx.consts
        energy, maps, *tensors = ctx.saved_tensors
        E = maps.shape[0]
        lib = _lib.lib()
        if ctx.plan.epoch != epoch:
            # the plan was refreshed since the forward pass (a setting re-assigned before backward): its pointer array may
            # address other tensors by now — the addresses of THIS graph come from the saved tensors
            ptrs = (ctypes.c_void_p * (E * MAX_PARAMS))()
            for pos, slots in enumerate(grad_slots):
                t = tensors[pos]
                for e, k, index in slots:
                    ptrs[e * MAX_PARAMS + k] = t.data_ptr() if index is None else t.data_ptr() + index * t.element_size()
        needs = ctx.needs_input_grad
        need_energy = needs[1]
        mask = [0] * E
        for pos, slots in enumerate(grad_slots):
            if needs[4 + pos]:
                for e, k, _ in slots:
                    mask[e] |= 1 << k
        if need_energy:
            mask = [m | (1 << MAX_PARAMS) for m in mask]
        ws_bytes = lib.chx_run_vjp_workspace_bytes(E)
        ws = workspace(ws_bytes, energy.device)
        dT = dT.to(energy.dtype).contiguous()
        d = torch.empty((E, MAX_PARAMS + 1), dtype=energy.dtype, device=energy.device)
        check(lib.chx_run_vjp_masked(kinds, ptrs, E, energy.data_ptr(), mass_eV, n_charges, dtype_code(energy.dtype), ptr(maps),
                                     ptr(dT), (ctypes.c_uint16 * E)(*mask), ptr(d), ptr(ws), ws_bytes, stream_ptr()),
              "chx_run_vjp_masked")
        grads = [None] * len(tensors)
        for pos, slots in enumerate(grad_slots):
            if not needs[4 + pos]:
                continue
            t = tensors[pos]
            if t.dim() == 0:
                g = d[slots[0][0], slots[0][1]]
                for e, k, _ in slots[1:]:
                    g = g + d[e, k]
            else:
                g = torch.zeros_like(t)
                for e, k, index in slots:
                    g[index] += d[e, k]
            grads[pos] = g
        d_energy = d[:, MAX_PARAMS].sum() if need_energy else None
        return (None, d_energy, None, None, *grads)


def build_compose_scalars(elements, energy: torch.Tensor, mass_eV: float, n_charges: float):
    """Composed (7,7) map of a run of elements whose builder parameters are all device scalars of the energy's dtype
    (chx_build_rmatrix_scalars + chx_compose_maps: two C calls whatever the number of elements; with gradients one autograd
    node, RunMapScalars). Returns None when the run does not qualify (vectorised or mixed-dtype parameters, elements without
    a builder kind) and the caller takes the general per-element path; without gradients the result is bit-identical either
    way."""
    if energy.dim() != 0 or not energy.is_cuda:
        return None
    tables = _scalar_run_tables(elements, energy)
    if tables is None:
        return None
    kinds, refs_all = tables
    if not kinds:
        return torch.eye(7, dtype=energy.dtype, device=energy.device)
    wants_grad = torch.is_grad_enabled() and (energy.requires_grad or any(t.requires_grad for refs in refs_all for t, _ in refs))
    if not wants_grad:
        with torch.no_grad():
            return _build_compose_raw(kinds, _scalar_run_pointers(refs_all), energy, mass_eV, n_charges)[1]
    tensors, where, slots = [], {}, []
    for refs in refs_all:
        row = []
        for t, index in refs:
            pos = where.get(id(t))
            if pos is None:
                pos = where[id(t)] = len(tensors)
                tensors.append(t)
            row.append((pos, index))
        slots.append(row)
    return RunMapScalars.apply((kinds, slots, mass_eV, n_charges), energy, *tensors)


COMPOSE_VJP_MAX = 192  # kComposeChunk: element maps per chx_compose_maps_vjp call


def _compose_raw(flat, bcast, B, dtype, device):
    E = len(flat)
    bc = (ctypes.c_uint8 * E)(*bcast)
    ptrs = (ctypes.c_void_p * E)(*[f.data_ptr() for f in flat])
    out = torch.empty((B, 7, 7), dtype=dtype, device=device)
    check(_lib.lib().chx_compose_maps(ptrs, bc, E, B, dtype_code(dtype), ptr(out), stream_ptr()), "chx_compose_maps")
    return out


class ComposeMaps(torch.autograd.Function):
    """R = M_E ... M_1 per batch row (chx_compose_maps); backward = chx_compose_maps_vjp (one wave per row, fp64 prefix /
    suffix sweep) — one autograd node for the whole product instead of a matmul node per element."""

    @staticmethod
    def forward(ctx, B, *flat):
        bcast = [1 if f.shape[0] == 1 else 0 for f in flat]
        ctx.save_for_backward(*flat)
        ctx.meta = (B, bcast)
        return _compose_raw(flat, bcast, B, flat[0].dtype, flat[0].device)

    @staticmethod
    def backward(ctx, dT):
        flat = ctx.saved_tensors
        B, bcast = ctx.meta
        E = len(flat)
        dtype, device = flat[0].dtype, flat[0].device
        lib = _lib.lib()
        dT = dT.to(dtype).contiguous()
        ws_bytes = lib.chx_compose_maps_vjp_workspace_bytes(E, B)
        ws = workspace(ws_bytes, device)
        dM = torch.empty((E, B, 7, 7), dtype=dtype, device=device)
        check(lib.chx_compose_maps_vjp((ctypes.c_void_p * E)(*[f.data_ptr() for f in flat]), (ctypes.c_uint8 * E)(*bcast), E, B,
                                       dtype_code(dtype), ptr(dT), ptr(dM), ptr(ws), ws_bytes, stream_ptr()), "chx_compose_maps_vjp")
        grads = []
        for e, f in enumerate(flat):
            if not ctx.needs_input_grad[1 + e]:
                grads.append(None)
            elif bcast[e] and B > 1:
                grads.append(dM[e].sum(dim=0, keepdim=True))
            else:
                grads.append(dM[e])
        return (None, *grads)


def compose_maps(maps: list[torch.Tensor], batch_shape, dtype, device) -> torch.Tensor:
    """maps: per-element (…,7,7) tensors -> composed (*batch_shape,7,7) = R_E … R_1 (segment.py:534-543)."""
    B = numel(batch_shape)
    flat = []
    for m in maps:
        f, Bm = flat_bcast(m if m.dtype == dtype else m.to(dtype), batch_shape, 2)
        flat.append(f.contiguous())
    if torch.is_grad_enabled() and any(f.requires_grad for f in flat):
        # longer products are folded in chunks of 192 maps (the kernel's by-value pointer table), each chunk one node
        out = None
        for lo in range(0, len(flat), COMPOSE_VJP_MAX - 1):
            chunk = flat[lo:lo + COMPOSE_VJP_MAX - 1]
            out = ComposeMaps.apply(B, *(chunk if out is None else [out] + chunk))
        return out.reshape(*batch_shape, 7, 7)
    out = _compose_raw(flat, [1 if f.shape[0] == 1 else 0 for f in flat], B, dtype, device)
    return out.reshape(*batch_shape, 7, 7)


# ---------------------------------------------------------------------------------------------
# linear apply
def _apply_raw(x, R, B, Bx, BR, N):
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_apply_affine7(ptr(x), ptr(R), ptr(out), B, Bx, BR, N, dtype_code(x.dtype), stream_ptr()),
          "chx_apply_affine7")
    return out


class Apply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, R, B):
        ctx.save_for_backward(x, R)
        ctx.B = B
        return _apply_raw(x, R, B, x.shape[0], R.shape[0], x.shape[1])

    @staticmethod
    def backward(ctx, dY):
        x, R = ctx.saved_tensors
        B, Bx, BR, N = ctx.B, x.shape[0], R.shape[0], x.shape[1]
        dY = aligned(dY)
        need_dx, need_dr = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        lib = _lib.lib()
        ws_bytes = lib.chx_apply_bwd_workspace_bytes(B, N)
        ws = workspace(ws_bytes, x.device)
        dX = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need_dx else None
        dR = torch.empty((B, 49), dtype=torch.float64, device=x.device) if need_dr else None
        check(lib.chx_apply_affine7_bwd(ptr(dY), ptr(R), ptr(x), ptr(dX), ptr(dR), B, Bx, BR, N,
                                        dtype_code(x.dtype), ptr(ws), ws_bytes, stream_ptr()),
              "chx_apply_affine7_bwd")
        if need_dx and Bx == 1 and B > 1:
            dX = dX.sum(dim=0, keepdim=True)
        if need_dr:
            dR = dR.reshape(B, 7, 7)
            if BR == 1 and B > 1:
                dR = dR.sum(dim=0, keepdim=True)
            dR = dR.to(R.dtype)
        return dX, dR, None


class _LinearSource:
    """Provenance of a tracked particle array y = R x whose only differentiable input is the MAP (the particles x carry no
    graph): what `moments` needs to differentiate a moment of y algebraically (mu' = A mu + b, cov' = A C A^T) instead of
    through two passes over the particles. Attached to the tensor object `apply_map` returns (`_chx_lin`); valid while that
    tensor's version counter stands still."""

    __slots__ = ("owner", "x", "R", "batch_shape", "version")

    def __init__(self, owner, x, R, batch_shape, version):
        self.owner, self.x, self.R, self.batch_shape, self.version = owner, x, R, batch_shape, version

    def rebound(self, copy_: torch.Tensor) -> "_LinearSource":
        return _LinearSource(self.owner, self.x, self.R, self.batch_shape, copy_._version)


def mark_copy_of(copy: torch.Tensor, source: torch.Tensor) -> None:
    """Declare `copy` an unmodified copy of `source` as both stand now (what clone_many records for its own copies): `_origin`
    follows the tag while neither tensor has been written to since."""
    root = _origin(source)
    copy._chx_origin = (weakref.ref(root), root._version, copy._version)


def _origin(t: torch.Tensor) -> torch.Tensor:
    """The tensor `t` is an unmodified copy of (clone_many), followed through chains of copies; `t` itself otherwise."""
    while True:
        tag = getattr(t, "_chx_origin", None)
        if tag is None or tag[2] != t._version:
            return t
        src = tag[0]()
        if src is None or src._version != tag[1]:
            return t
        t = src


def apply_map(particles: torch.Tensor, tm: torch.Tensor) -> torch.Tensor:
    """`particles @ tm.mT` (element.py:182) with torch broadcasting of the vector dims."""
    require_device(particles, tm)
    if tm.dtype != particles.dtype:
        raise RuntimeError(f"transfer map dtype {tm.dtype} does not match particle dtype {particles.dtype}")
    N = particles.shape[-2]
    batch_shape = bshapes(particles.shape[:-2], tm.shape[:-2])
    B = numel(batch_shape)
    x, _ = flat_bcast(particles, batch_shape, 2)
    R, _ = flat_bcast(tm, batch_shape, 2)
    x, R = aligned(x), R.contiguous()
    if x.requires_grad or R.requires_grad:
        out = Apply.apply(x, R, B).reshape(*batch_shape, N, 7)
        if not x.requires_grad and torch.is_grad_enabled():
            out._chx_lin = _LinearSource(_origin(particles), x, R, batch_shape, out._version)
        return out
    out = _apply_raw(x, R, B, x.shape[0], R.shape[0], N)
    return out.reshape(*batch_shape, N, 7)


def track_elementwise(particles, maps: torch.Tensor, fused: bool = False) -> torch.Tensor:
    """Apply E maps ([E][BR][7][7]) one after the other without merging them."""
    require_device(particles, maps)
    if torch.is_grad_enabled() and (particles.requires_grad or maps.requires_grad):
        # gradient path: E differentiable passes (Apply); the one-launch kernels below keep no intermediates
        x = particles
        for e in range(maps.shape[0]):
            x = apply_map(x, maps[e] if maps.shape[1] > 1 else maps[e, 0])
        return x
    E, BR = maps.shape[0], maps.shape[1]
    N = particles.shape[-2]
    batch_shape = torch.broadcast_shapes(particles.shape[:-2], (BR,) if BR > 1 else ())
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    x, maps = aligned(x), maps.contiguous()
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    lib = _lib.lib()
    if fused:
        check(lib.chx_track_fused(ptr(x), ptr(maps), ptr(out), E, B, Bx, BR, N, dtype_code(x.dtype), stream_ptr()),
              "chx_track_fused")
    else:
        check(lib.chx_track_elementwise(ptr(x), ptr(maps), ptr(out), None, E, B, Bx, BR, N,
                                        dtype_code(x.dtype), stream_ptr()), "chx_track_elementwise")
    return out.reshape(*batch_shape, N, 7)


# ---------------------------------------------------------------------------------------------
# cavity
def _cavity_coeffs_autograd(params, energy, mass_eV, n_charges, B):
    """The closed forms of chx_cavity_coeffs (cavity.py:113-122,135-226) as (B,)-sized fp64 tensor expressions, used
    only when voltage / phase / frequency / length / energy carry gradients: autograd differentiates them, the
    particle-sized work stays in the kernels (CavityTrack)."""
    import math

    p = params.to(torch.float64).expand(B, 4)
    L, V, phi, freq = p[:, 0], p[:, 1], p[:, 2] * (math.pi / 180.0), p[:, 3]
    E0 = energy.to(torch.float64).expand(B)
    g0 = E0 / mass_eV
    ig2 = 1.0 / (g0 * g0)
    b0 = (1.0 - ig2).sqrt()
    cphi, sphi = phi.cos(), phi.sin()
    dEn = V * cphi * n_charges * -1.0
    E1 = E0 + dEn
    g1 = E1 / mass_eV
    b1 = (1.0 - 1.0 / (g1 * g1)).sqrt()
    k = 2.0 * math.pi * freq / 299792458.0
    gain = (dEn > 0).any()
    dg = V / mass_eV
    b03, b13, g03, g13 = b0**3, b1**3, g0**3, g1**3
    gd = g0 - g1
    gd = torch.where(gd == 0, torch.ones_like(gd), gd)  # only reached in rows the `gain` branch does not use
    T566_gain = L * (b03 * g03 - b13 * g13) / (2.0 * b0 * b13 * g0 * gd * g13)
    T556_gain = b0 * k * L * dg * g0 * (b13 * g13 + b0 * (g0 - g13)) * sphi / (b13 * g13 * gd * gd)
    T555_gain = b0 * b0 * k * k * L * dg / 2.0 * (
        dg * (2.0 * g0 * g13 * (b0 * b13 - 1.0) + g0 * g0 + 3.0 * g1 * g1 - 2.0) / (b13 * g13 * gd**3) * sphi * sphi
        - (g1 * g0 * (b1 * b0 - 1.0) + 1.0) / (b1 * g1 * gd * gd) * cphi)
    zero = torch.zeros_like(L)
    T566 = torch.where(gain, T566_gain, 1.5 * L * ig2 / b03)
    T556 = torch.where(gain, T556_gain, zero)
    T555 = torch.where(gain, T555_gain, zero)
    coeffs = torch.stack([E0 * b0 / (E1 * b1), V * b0 / (E1 * b1), b0 * k, phi, cphi, T566, T556, T555], dim=-1)
    return coeffs, E1.to(energy.dtype)


def cavity_coeffs(params, energy, mass_eV, n_charges, B):
    if params.requires_grad or energy.requires_grad:
        return _cavity_coeffs_autograd(params, energy, mass_eV, n_charges, B)
    coeffs = torch.empty((B, CAV_NCOEF), dtype=torch.float64, device=energy.device)
    e_out = torch.empty((B,), dtype=energy.dtype, device=energy.device)
    check(_lib.lib().chx_cavity_coeffs(ptr(params), ptr(energy), mass_eV, n_charges, B, params.shape[0],
                                       energy.shape[0], dtype_code(energy.dtype), ptr(coeffs), ptr(e_out),
                                       stream_ptr()), "chx_cavity_coeffs")
    return coeffs, e_out


def _cavity_track_raw(x, R, coeffs, B, N):
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_cavity_track(ptr(x), ptr(R), ptr(coeffs), ptr(out), B, x.shape[0], N,
                                      dtype_code(x.dtype), stream_ptr()), "chx_cavity_track")
    return out


class CavityTrack(torch.autograd.Function):
    """y = chx_cavity_track(x, R, coeffs); backward = chx_apply_affine7_bwd on the matrix part (row 5 of R does not
    reach the output: delta' is rewritten from the incoming tau, delta) + chx_cavity_track_bwd for the rewrite."""

    @staticmethod
    def forward(ctx, x, R, coeffs, B):
        ctx.save_for_backward(x, R, coeffs)
        ctx.B = B
        return _cavity_track_raw(x, R, coeffs, B, x.shape[1])

    @staticmethod
    def backward(ctx, dY):
        x, R, coeffs = ctx.saved_tensors
        B, Bx, N = ctx.B, x.shape[0], x.shape[1]
        dY = aligned(dY)
        need_dx, need_dr, need_dc = ctx.needs_input_grad[:3]
        lib = _lib.lib()
        dt = dtype_code(x.dtype)
        dX = dR = None
        if need_dx or need_dr:
            R0 = R.clone()
            R0[:, 5, :] = 0
            ws_bytes = lib.chx_apply_bwd_workspace_bytes(B, N)
            ws = workspace(ws_bytes, x.device)
            dX = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need_dx else None
            dR = torch.empty((B, 49), dtype=torch.float64, device=x.device) if need_dr else None
            check(lib.chx_apply_affine7_bwd(ptr(dY), ptr(R0), ptr(x), ptr(dX), ptr(dR), B, Bx, B, N, dt, ptr(ws),
                                            ws_bytes, stream_ptr()), "chx_apply_affine7_bwd")
        dC = None
        if need_dx or need_dc:
            ws_bytes = lib.chx_moments_workspace_bytes(B, N)
            ws = workspace(ws_bytes, x.device)
            dC = torch.empty((B, CAV_NCOEF), dtype=torch.float64, device=x.device)
            check(lib.chx_cavity_track_bwd(ptr(dY), ptr(x), ptr(coeffs), ptr(dX), ptr(dC), B, Bx, N, dt, ptr(ws), ws_bytes,
                                           stream_ptr()), "chx_cavity_track_bwd")
        if need_dx and Bx == 1 and B > 1:
            dX = dX.sum(dim=0, keepdim=True)
        if need_dr:
            dR = dR.reshape(B, 7, 7).clone()
            dR[:, 5, :] = 0
            dR = dR.to(R.dtype)
        return dX, dR, (dC if need_dc else None), None


def cavity_track(x, R, coeffs, B, N):
    if x.requires_grad or R.requires_grad or coeffs.requires_grad:
        return CavityTrack.apply(x, R, coeffs.contiguous(), B)
    return _cavity_track_raw(x, R, coeffs, B, N)


# ---------------------------------------------------------------------------------------------
# non-linear tracking (drift_kick_drift, second_order)
DKD_KIND = {"drift": 0, "quadrupole": 1, "dipole": 2, "tdc": 3}
DKD_NUM_PARAMS = [1, 5, 9, 7]
T_KIND = {"drift": 0, "quadrupole": 1, "dipole": 2, "sextupole": 3, "general": 4}
T_NUM_PARAMS = [1, 5, 9, 5, 4]
FRINGE_AT = {"neither": 0, "entrance": 1, "exit": 2, "both": 3}


def stack_params(values, dtype, device) -> tuple[torch.Tensor, torch.Size]:
    """Element parameters (0-d or vector tensors) -> ((Bp, P) tensor, their common vector shape)."""
    shape = bshapes(*[v.shape for v in values])
    if len(shape) == 0:
        return torch.stack([v.to(dtype) for v in values]).reshape(1, len(values)), shape
    return torch.stack([v.to(dtype).expand(shape) for v in values], dim=-1).reshape(-1, len(values)), shape


#: `Element.dkd_precision` -> the `storage_precision` code of chx_dkd_track_p (0 float64 arithmetic, 1 float32, 2 mixed)
DKD_PRECISION = {"double": 0, "storage": 1, "mixed": 2}


def _dkd_raw(kind, x, p, e, mass_eV, n_charges, num_steps, fringe_at, B, N, storage_precision=0):
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    e_out = torch.empty((B,), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_dkd_track_p(kind, ptr(x), ptr(p), ptr(e), mass_eV, n_charges, num_steps, fringe_at, B, x.shape[0],
                                     p.shape[0], e.shape[0], N, dtype_code(x.dtype), int(storage_precision), ptr(out),
                                     ptr(e_out), stream_ptr()), "chx_dkd_track_p")
    return out, e_out


DKD_LINEAR = 4      # chx_dkd_chain_mixed: a merged run of linear elements between drift-kick-drift elements


def dkd_chain(kinds, params, num_steps, fringe, storage, x, energy, s, mass_eV, n_charges, arrays=None, lengths=None):
    """A run of drift-kick-drift elements on one plain beam (chx_dkd_chain_mixed): x (N, 7), energy and s 0-d of x's dtype, params
    = the elements' (1, P) parameter arrays. Items of kind DKD_LINEAR are merged runs of linear elements in between: their
    `params` entry is the run's (7, 7) map, their `lengths` entry its 0-d summed length (None for the other items). Returns
    (particles (N, 7), energy 0-d, s 0-d, arrays) behind the last item; `arrays` (the argument arrays) can be handed back in
    while the run and its tensors are the same."""
    E, N = len(kinds), x.shape[0]
    x = aligned(x)
    out = torch.empty((N, 7), dtype=x.dtype, device=x.device)
    tmp = torch.empty((N, 7), dtype=x.dtype, device=x.device) if E > 1 else None
    scalars = torch.empty((E + 1,), dtype=x.dtype, device=x.device)       # the elements' outgoing energies, then s
    if arrays is None:
        i32 = ctypes.c_int32 * E
        vp = ctypes.c_void_p * E
        lens = vp(*[None if t is None else t.data_ptr() for t in lengths]) if lengths is not None else None
        arrays = (i32(*kinds), vp(*[p.data_ptr() for p in params]), i32(*num_steps), i32(*fringe), i32(*storage), lens)
    check(_lib.lib().chx_dkd_chain_mixed(arrays[0], arrays[1], arrays[5], arrays[2], arrays[3], arrays[4], E, ptr(x), ptr(energy), mass_eV,
                                         n_charges, N, dtype_code(x.dtype), ptr(out), ptr(tmp), ptr(scalars), ptr(s),
                                         scalars.data_ptr() + E * scalars.element_size(), stream_ptr()), "chx_dkd_chain_mixed")
    return out, scalars[E - 1], scalars[E], arrays


def dkd_energy_chain(kinds, energy, mass_eV):
    """(E,) reference energies behind the items of a drift-kick-drift run (chx_dkd_energy_chain): a drift-kick-drift element
    leaves the round trip of what it received in the beam's dtype (bmadx.py:49), a DKD_LINEAR item hands its energy on."""
    E = len(kinds)
    out = torch.empty((E,), dtype=energy.dtype, device=energy.device)
    scratch = torch.empty((E,), dtype=torch.int32, device=energy.device)
    check(_lib.lib().chx_dkd_energy_chain((ctypes.c_int32 * E)(*kinds), E, ptr(energy), mass_eV, dtype_code(energy.dtype), ptr(out),
                                          ptr(scratch), stream_ptr()), "chx_dkd_energy_chain")
    return out


class DkdTrack(torch.autograd.Function):
    """(x_out, ref_energy) = chx_dkd_track(x, params, energy); backward = chx_dkd_track_bwd (dual numbers on device)."""

    @staticmethod
    def forward(ctx, x, p, e, kind, mass_eV, n_charges, num_steps, fringe_at, B, N):
        ctx.save_for_backward(x, p, e)
        ctx.meta = (kind, mass_eV, n_charges, num_steps, fringe_at, B, N)
        return _dkd_raw(kind, x, p, e, mass_eV, n_charges, num_steps, fringe_at, B, N)

    @staticmethod
    def backward(ctx, dY, dE):
        x, p, e = ctx.saved_tensors
        kind, mass_eV, n_charges, num_steps, fringe_at, B, N = ctx.meta
        P = DKD_NUM_PARAMS[kind]
        need_x = ctx.needs_input_grad[0]
        need_theta = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dY = aligned(dY.contiguous())
        dx = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need_x else None
        partials = None
        if need_theta:
            count = _lib.lib().chx_dkd_bwd_partials_count(kind, B, N)
            partials = torch.empty((count,), dtype=torch.float64, device=x.device)
        if need_x or need_theta:
            check(_lib.lib().chx_dkd_track_bwd(kind, ptr(x), ptr(p), ptr(e), ptr(dY), mass_eV, n_charges, num_steps,
                                               fringe_at, B, x.shape[0], p.shape[0], e.shape[0], N, dtype_code(x.dtype),
                                               ptr(dx), ptr(partials), stream_ptr()), "chx_dkd_track_bwd")
        dp = de = None
        if need_theta:
            tot = partials.view(B, -1, P + 1).sum(dim=1)  # (B, P + 1) fp64
            if ctx.needs_input_grad[1]:
                dp = tot[:, :P]
                dp = (dp.sum(dim=0, keepdim=True) if p.shape[0] == 1 and B > 1 else dp).to(p.dtype)
            if ctx.needs_input_grad[2]:
                de = tot[:, P] + dE.to(torch.float64)  # d ref_energy / d energy = 1 (bmadx.py:49)
                de = (de.sum(dim=0, keepdim=True) if e.shape[0] == 1 and B > 1 else de).to(e.dtype)
        if need_x and x.shape[0] == 1 and B > 1:
            dx = dx.sum(dim=0, keepdim=True)
        return dx, dp, de, None, None, None, None, None, None, None


def dkd_track(kind: int, particles, params, param_shape, energy, mass_eV: float, n_charges: float, num_steps: int = 1,
              fringe_at: int = 3, storage_precision: int = 0):
    """One drift-kick-drift element (chx_dkd_track_p): particles (..., N, 7), params (Bp, P) with vector shape
    `param_shape`, energy (...). Returns (particles_out (*batch, N, 7), ref_energy (*energy/param batch)).
    `storage_precision` (DKD_PRECISION): float32 beams in float64 arithmetic (0), in float32 like the reference's own tensor code
    (1), or mixed (2: the longitudinal pair in float64, the rest in float32; Drift and Quadrupole); the differentiable path always
    runs in float64 (dual numbers)."""
    require_device(particles, params, energy)
    N = particles.shape[-2]
    eb_shape = bshapes(param_shape, energy.shape)           # batch shape of the outgoing energy
    batch_shape = bshapes(particles.shape[:-2], eb_shape)
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    p, Bp = flat_bcast(params.reshape(*param_shape, params.shape[-1]), batch_shape, 1)
    e, Be = flat_bcast(energy, batch_shape, 0)
    x, p, e = aligned(x), p.contiguous(), e.contiguous()
    if torch.is_grad_enabled() and (x.requires_grad or p.requires_grad or e.requires_grad):
        out, e_out = DkdTrack.apply(x, p, e, kind, mass_eV, n_charges, num_steps, fringe_at, B, N)
    else:
        out, e_out = _dkd_raw(kind, x, p, e, mass_eV, n_charges, num_steps, fringe_at, B, N, storage_precision)
    # the outgoing reference energy has the INCOMING energy's shape (bmadx.py:49: it only depends on p0c)
    if energy.numel() == 1:
        e_out = e_out[:1].reshape(energy.shape)
    elif tuple(energy.shape) == tuple(batch_shape):
        e_out = e_out.reshape(batch_shape)
    else:  # energy broadcast against other vector dims: pick one representative per energy entry
        idx = torch.arange(energy.numel(), device=x.device).reshape(energy.shape).expand(batch_shape).reshape(-1)
        e_out = torch.zeros(energy.numel(), dtype=x.dtype, device=x.device).index_copy(0, idx, e_out).reshape(energy.shape)
    return out.reshape(*batch_shape, N, 7), e_out


def _build_ttensor_raw(kind, p, e, mass_eV, B):
    T = torch.empty((B, 7, 7, 7), dtype=e.dtype, device=e.device)
    check(_lib.lib().chx_build_ttensor(kind, ptr(p), ptr(e), mass_eV, B, p.shape[0], e.shape[0], dtype_code(e.dtype), ptr(T),
                                       stream_ptr()), "chx_build_ttensor")
    return T


class BuildTTensor(torch.autograd.Function):
    """T = chx_build_ttensor(params, energy); backward = chx_build_ttensor_vjp (dual numbers on device)."""

    @staticmethod
    def forward(ctx, p, e, kind, mass_eV, B):
        ctx.save_for_backward(p, e)
        ctx.meta = (kind, mass_eV, B)
        return _build_ttensor_raw(kind, p, e, mass_eV, B)

    @staticmethod
    def backward(ctx, dT):
        p, e = ctx.saved_tensors
        kind, mass_eV, B = ctx.meta
        P = T_NUM_PARAMS[kind]
        dT = dT.contiguous()
        dp = torch.empty((B, P), dtype=e.dtype, device=e.device)
        de = torch.empty((B,), dtype=e.dtype, device=e.device)
        check(_lib.lib().chx_build_ttensor_vjp(kind, ptr(p), ptr(e), mass_eV, ptr(dT), B, p.shape[0], e.shape[0],
                                               dtype_code(e.dtype), ptr(dp), ptr(de), stream_ptr()), "chx_build_ttensor_vjp")
        if p.shape[0] == 1 and B > 1:
            dp = dp.sum(dim=0, keepdim=True)
        if e.shape[0] == 1 and B > 1:
            de = de.sum(dim=0, keepdim=True)
        return dp, de, None, None, None


def build_ttensor(kind: int, params, param_shape, energy, mass_eV: float) -> torch.Tensor:
    """Second-order transfer tensor (*batch, 7, 7, 7) of one element (chx_build_ttensor)."""
    require_device(params, energy)
    batch_shape = bshapes(param_shape, energy.shape)
    B = max(numel(batch_shape), 1)
    p, Bp = flat_bcast(params.reshape(*param_shape, params.shape[-1]), batch_shape, 1)
    e, Be = flat_bcast(energy, batch_shape, 0)
    p, e = p.contiguous(), e.contiguous()
    if torch.is_grad_enabled() and (p.requires_grad or e.requires_grad):
        T = BuildTTensor.apply(p, e, kind, mass_eV, B)
    else:
        T = _build_ttensor_raw(kind, p, e, mass_eV, B)
    return T.reshape(*batch_shape, 7, 7, 7)


def _apply_second_order_raw(x, Tt, B, N):
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_apply_second_order(ptr(x), ptr(Tt), ptr(out), B, x.shape[0], Tt.shape[0], N, dtype_code(x.dtype),
                                            stream_ptr()), "chx_apply_second_order")
    return out


_UNFOLD_INDEX: dict = {}


def _unfold_index(device) -> torch.Tensor:
    """(343,) index of (i, j, k) into the folded (i, j <= k) order of chx_apply_second_order_bwd."""
    idx = _UNFOLD_INDEX.get(device)
    if idx is None:
        start = [sum(7 - a for a in range(j)) for j in range(7)]
        flat = [i * 28 + start[min(j, k)] + abs(k - j) for i in range(7) for j in range(7) for k in range(7)]
        idx = _UNFOLD_INDEX[device] = torch.tensor(flat, dtype=torch.int64, device=device)
    return idx


class ApplySecondOrder(torch.autograd.Function):
    """y = T x x (chx_apply_second_order); backward = chx_apply_second_order_bwd."""

    @staticmethod
    def forward(ctx, x, Tt, B, N):
        ctx.save_for_backward(x, Tt)
        ctx.meta = (B, N)
        return _apply_second_order_raw(x, Tt, B, N)

    @staticmethod
    def backward(ctx, dY):
        x, Tt = ctx.saved_tensors
        B, N = ctx.meta
        need_x, need_T = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dY = aligned(dY.contiguous())
        dx = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need_x else None
        partials = None
        if need_T:
            count = _lib.lib().chx_second_order_bwd_partials_count(B)
            partials = torch.empty((count,), dtype=torch.float64, device=x.device)
        check(_lib.lib().chx_apply_second_order_bwd(ptr(x), ptr(Tt), ptr(dY), ptr(dx), ptr(partials), B, x.shape[0],
                                                    Tt.shape[0], N, dtype_code(x.dtype), stream_ptr()),
              "chx_apply_second_order_bwd")
        dT = None
        if need_T:
            dU = partials.view(B, -1, 196).sum(dim=1)
            dT = dU.index_select(1, _unfold_index(x.device))
            dT = (dT.sum(dim=0, keepdim=True) if Tt.shape[0] == 1 and B > 1 else dT).to(Tt.dtype).reshape(-1, 7, 7, 7)
        if need_x and x.shape[0] == 1 and B > 1:
            dx = dx.sum(dim=0, keepdim=True)
        return dx, dT, None, None


def apply_second_order(particles: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """einsum("...ijk,...j,...k->...i", T, x, x) over the particle axis (element.py:211-216)."""
    require_device(particles, T)
    if T.dtype != particles.dtype:
        raise RuntimeError(f"transfer map dtype {T.dtype} does not match particle dtype {particles.dtype}")
    N = particles.shape[-2]
    batch_shape = bshapes(particles.shape[:-2], T.shape[:-3])
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    Tt, BT = flat_bcast(T, batch_shape, 3)
    x, Tt = aligned(x), Tt.contiguous()
    if torch.is_grad_enabled() and (x.requires_grad or Tt.requires_grad):
        out = ApplySecondOrder.apply(x, Tt, B, N)
    else:
        out = _apply_second_order_raw(x, Tt, B, N)
    return out.reshape(*batch_shape, N, 7)


def second_order_chain(T_maps, lengths, x, s, ptrs=None, linear=None):
    """A run of second-order elements on one plain beam (chx_second_order_chain_mixed): x (N, 7); T_maps = the elements' (7, 7, 7)
    maps, lengths their 0-d length tensors, s the 0-d path length — all of x's dtype. `linear` (a list of 0 / 1, optional) marks
    entries of T_maps that are (7, 7) first-order maps of merged linear runs in between (their `lengths` entry: the run's summed
    length). Returns (particles (N, 7), s, ptrs); `ptrs` (the argument arrays) can be handed back in while T_maps and lengths are
    the same tensors."""
    E, N = len(T_maps), x.shape[0]
    x = aligned(x)
    out = torch.empty((N, 7), dtype=x.dtype, device=x.device)
    tmp = torch.empty((N, 7), dtype=x.dtype, device=x.device) if E > 1 else None
    s_out = torch.empty((), dtype=x.dtype, device=x.device)
    if ptrs is None:
        arr = ctypes.c_void_p * E
        flags = (ctypes.c_int32 * E)(*linear) if linear is not None and any(linear) else None
        ptrs = (arr(*[t.data_ptr() for t in T_maps]), arr(*[t.data_ptr() for t in lengths]), flags)
    check(_lib.lib().chx_second_order_chain_mixed(ptrs[0], ptrs[2], ptrs[1], E, ptr(x), N, dtype_code(x.dtype), ptr(out), ptr(tmp),
                                                  ptr(s), ptr(s_out), stream_ptr()), "chx_second_order_chain_mixed")
    return out, s_out, ptrs


# ---------------------------------------------------------------------------------------------
# moments
#: the batch (vector) index of the reductions and deposits is `blockIdx.y`: at most 65 535 rows per launch
MAX_GRID_ROWS = 65535


def _moments_raw(x, w, B, N, entry=None):
    """chx_moments; with entry = (index, take_sqrt) also that entry of every row in x's dtype (chx_moments_entry: the same two
    launches) -> (moments, entries)."""
    lib = _lib.lib()
    if B > MAX_GRID_ROWS and N > 2048:     # (short rows: one wave per row, the rows are blockIdx.x)
        # more vector rows than one launch takes (the batch index is a grid dimension of 65 535): row slices, one call each
        outs = []
        for b0 in range(0, B, MAX_GRID_ROWS):
            b1 = min(B, b0 + MAX_GRID_ROWS)
            outs.append(_moments_raw(x if x.shape[0] == 1 else x[b0:b1], w if w is None or w.shape[0] == 1 else w[b0:b1],
                                     b1 - b0, N, entry))
        if entry is None:
            return torch.cat(outs)
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    ws_bytes = lib.chx_moments_workspace_bytes(B, N)
    ws = workspace(ws_bytes, x.device)
    out = torch.empty((B, MOM_NOUT), dtype=torch.float64, device=x.device)
    if entry is None:
        check(lib.chx_moments(ptr(x), ptr(w), B, x.shape[0], 1 if w is None else w.shape[0], N, dtype_code(x.dtype),
                              ptr(out), ptr(ws), ws_bytes, stream_ptr()), "chx_moments")
        return out
    picked = torch.empty((B,), dtype=x.dtype, device=x.device)
    check(lib.chx_moments_entry(ptr(x), ptr(w), B, x.shape[0], 1 if w is None else w.shape[0], N, dtype_code(x.dtype), ptr(out),
                                entry[0], int(entry[1]), ptr(picked), ptr(ws), ws_bytes, stream_ptr()), "chx_moments_entry")
    return out, picked


def _moments_bwd_raw(x, w, out, d_out, B, need_x, need_w):
    """chx_moments_bwd_w: (dX | None, dW | None) of d_out . moments for the rows x (Bx, N, 7) / weights w (Bw, N) given the
    moment vector `out` the cotangent belongs to. The kernel forms every term from `out` (W, W2, mu, cov) and the row itself, so
    with the moments of ALL shards of a particle-sharded beam in `out` it returns the gradient of the GLOBAL statistics with
    respect to this rank's rows (utils/statistics.py:4-62 over the union of the shards)."""
    N = x.shape[1]
    dX = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need_x else None
    dW = torch.empty((B, N), dtype=x.dtype, device=x.device) if need_w else None
    check(_lib.lib().chx_moments_bwd_w(ptr(x), ptr(w), ptr(out), ptr(d_out), B, x.shape[0], 1 if w is None else w.shape[0], N,
                                       dtype_code(x.dtype), ptr(dX), ptr(dW), stream_ptr()), "chx_moments_bwd_w")
    return dX, dW


class Moments(torch.autograd.Function):
    """out = chx_moments(x, w); backward = chx_moments_bwd_w: gradients of the particles AND of the weights (the reference's
    weighted statistics are differentiable in both, utils/statistics.py:4-62) in one pass.

    With a process `group` (a particle-sharded beam, `sharding.particle_sharded`): forward = this rank's one-pass moments, ONE
    all-gather of 29 doubles per rank and row, the exact merge — the statistics of the union of the shards; backward = the same
    kernel on THIS rank's rows with the GLOBAL moments: no collective. Every rank evaluates the same loss of the global
    statistics, so what a rank's backward pass accumulates on a replicated setting is its shard's share of the gradient:
    `sharding.all_reduce_gradients` (or any data-parallel wrapper) sums the shares."""

    @staticmethod
    def forward(ctx, x, w, B, group=None):
        out = _moments_raw(x, w, B, x.shape[1])
        if group is not None:
            from . import sharding

            out = sharding.gather_merge_moments(out, group)
        ctx.save_for_backward(x, w, out)
        ctx.B = B
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w, out = ctx.saved_tensors
        B = ctx.B
        need_x, need_w = ctx.needs_input_grad[0], w is not None and ctx.needs_input_grad[1]
        d_out = d_out.contiguous().to(torch.float64)
        dX, dW = _moments_bwd_raw(x, w, out, d_out, B, need_x, need_w)
        if need_x and x.shape[0] == 1 and B > 1:
            dX = dX.sum(dim=0, keepdim=True)
        if need_w and w.shape[0] == 1 and B > 1:
            dW = dW.sum(dim=0, keepdim=True)
        return dX, dW, None, None


def _memo_moments(owner: torch.Tensor, x: torch.Tensor, w, survival, B: int, entry=None):
    """chx_moments (B,29) of the flat, gradient-free x (Bx,N,7) / w (Bw,N), memoised on the tensor OBJECT `owner` the values
    belong to, against (its version, the weight tensor — followed to the array it is an unmodified copy of — and that
    array's version, the data address). An optimisation loop tracks one incoming beam again and again: reduced once.
    With entry = (index, take_sqrt): returns (moments, that entry per row in x's dtype — or None on a memo hit, the caller then
    picks it with chx_moment_entry)."""
    w_src = None if survival is None else _origin(survival)
    cached = getattr(owner, "_chx_mom", None) if (not CAPTURING[0] or CAPTURE_KEEPS_BEAM_MOMENTS[0]) else None
    if cached is not None and cached[0] == owner._version and cached[1] is w_src \
            and (w_src is None or cached[2] == w_src._version) and cached[3] == x.data_ptr() and cached[4].shape[0] == B:
        return cached[4] if entry is None else (cached[4], None)
    res = _moments_raw(x, w, B, x.shape[1], entry)
    mom = res if entry is None else res[0]
    owner._chx_mom = (owner._version, w_src, None if w_src is None else w_src._version, x.data_ptr(), mom)
    return res


def _incoming_moments(lin: "_LinearSource", survival, w, B: int) -> torch.Tensor:
    """chx_moments (Bm,29) of the beam that ENTERED the linear map, with the weights the outgoing moment was taken with."""
    x = lin.x
    return _memo_moments(lin.owner, x, w, survival, max(x.shape[0], 1 if w is None else w.shape[0]))


class MomentsMapped(torch.autograd.Function):
    """out = chx_moments(y) for y = R x with x free of gradients: forward reduces the tracked particles like `Moments`
    (same numbers), backward = chx_moments_mapped_bwd — 7x7 algebra on the INCOMING beam's moments, no pass over y or x."""

    @staticmethod
    def forward(ctx, R, y, w, source, B):
        out = _moments_raw(y, w, B, y.shape[1])
        ctx.save_for_backward(R, source[0].x, *(() if w is None else (w,)))
        ctx.meta = (source, B)
        return out

    @staticmethod
    def backward(ctx, d_out):
        R, x, *rest = ctx.saved_tensors
        (lin, survival), B = ctx.meta
        w = rest[0] if rest else None
        mom_x = _incoming_moments(lin, survival, w, B)
        d_out = d_out.contiguous().to(torch.float64)
        dR = torch.empty((B, 49), dtype=torch.float64, device=R.device)
        check(_lib.lib().chx_moments_mapped_bwd(ptr(d_out), ptr(R), ptr(mom_x), B, R.shape[0], mom_x.shape[0],
                                                dtype_code(R.dtype), ptr(dR), stream_ptr()), "chx_moments_mapped_bwd")
        dR = dR.reshape(B, 7, 7)
        if R.shape[0] == 1 and B > 1:
            dR = dR.sum(dim=0, keepdim=True)
        return dR.to(R.dtype), None, None, None, None


class MomentEntryMapped(torch.autograd.Function):
    """One beam property (mu_*, sigma_*, cov_*) of a linearly tracked beam as ONE autograd node hanging on the map:
    forward = chx_moment_entry on the (memoised) moment vector of the tracked particles, backward = chx_moment_entry_mapped_bwd
    (scalar gradient -> dR in one launch, from the incoming beam's memoised moments)."""

    @staticmethod
    def forward(ctx, R, mom_y, source, index, take_sqrt, B, picked=None):
        if picked is not None:                      # came with the reduction (chx_moments_entry)
            out = picked
        else:
            out = torch.empty((B,), dtype=R.dtype, device=R.device)
            check(_lib.lib().chx_moment_entry(ptr(mom_y), B, index, int(take_sqrt), dtype_code(R.dtype), ptr(out), stream_ptr()),
                  "chx_moment_entry")
        lin, survival, w = source
        ctx.save_for_backward(R, mom_y, lin.x, *(() if w is None else (w,)))
        ctx.meta = (lin, survival, index, take_sqrt, B)
        return out

    @staticmethod
    def backward(ctx, grad):
        R, mom_y, x, *rest = ctx.saved_tensors
        lin, survival, index, take_sqrt, B = ctx.meta
        mom_x = _incoming_moments(lin, survival, rest[0] if rest else None, B)
        grad = grad.contiguous()
        BR = R.shape[0]
        direct = BR == B
        dR = torch.empty((B, 7, 7), dtype=R.dtype if direct else torch.float64, device=R.device)
        check(_lib.lib().chx_moment_entry_mapped_bwd(ptr(grad), ptr(mom_y), index, int(take_sqrt), ptr(R), ptr(mom_x), B, BR,
                                                     mom_x.shape[0], dtype_code(R.dtype), ptr(dR), 0 if direct else 1,
                                                     stream_ptr()), "chx_moment_entry_mapped_bwd")
        if not direct:
            dR = dR.sum(dim=0, keepdim=True).to(R.dtype)
        return dR, None, None, None, None, None, None


#: a property of a [run | Screen] stretch's record as ONE node on the run's settings (False: on the run's composed map, whose own node
#: carries the gradient on to the settings — the two-node form; tests compare the two)
ONE_NODE_PROPERTY = [True]


def moment_entry(particles: torch.Tensor, survival, index: int, take_sqrt: bool):
    """`moments(particles, survival)[..., index]` (or its square root) in the particle dtype for a linearly tracked beam
    whose particles carry no graph of their own (`_LinearSource`), as one autograd node on the map; None when that does not
    apply and the caller takes `moments`."""
    lin = getattr(particles, "_chx_lin", None)
    if lin is None or lin.version != particles._version or not lin.R.requires_grad or not torch.is_grad_enabled():
        return None
    if survival is not None and survival.requires_grad:
        return None
    if particles.dim() == 2 and lin.batch_shape == () and lin.R.shape[0] == 1 and lin.R.dtype == particles.dtype and lin.x.shape[0] == 1 \
            and (survival is None or (survival.dim() == 1 and survival.dtype == particles.dtype and survival.is_contiguous())) \
            and particles.is_contiguous() and particles.data_ptr() % 16 == 0:
        # the plain case — one beam through one map, an optimisation loop's every step — without the broadcasting preparations:
        # the memoised moments of the incoming beam, the node in C++ (cheetah_amd._chxtorch MomentEntryMappedNode)
        memo = not CAPTURING[0] or CAPTURE_KEEPS_BEAM_MOMENTS[0]
        w_src = None if survival is None else _origin(survival)
        w_ver = None if w_src is None else w_src._version
        owner, mom_x = lin.owner, None
        cached = getattr(owner, "_chx_mom", None) if memo else None
        if cached is not None and cached[0] == owner._version and cached[1] is w_src and cached[2] == w_ver \
                and cached[3] == lin.x.data_ptr() and cached[4].shape[0] == 1:
            mom_x = cached[4]
        else:
            mom_x = _memo_moments(owner, lin.x, None if survival is None else survival.reshape(1, -1), survival, 1)
        cached = getattr(particles, "_chx_mom", None) if memo else None
        known = None
        if cached is not None and cached[0] == particles._version and cached[1] is w_src and cached[2] == w_ver \
                and cached[3] == particles.data_ptr() and cached[4].shape[0] == 1:
            known = cached[4]
        # (sums, version of the rows, the weights they were taken with, the run the rows came through): what the particle pass of a
        # differentiable [run | Screen] stretch left on its record
        tag = getattr(particles, "_chx_partials", None)
        partials = tag[0] if tag is not None and tag[1] == particles._version and tag[2] is survival else None
        run = tag[3] if partials is not None and len(tag) > 3 else None
        if run is not None and run[0] is lin.R and index >= 2 and ONE_NODE_PROPERTY[0]:
            # the rows are the record of a [run | Screen] stretch: the property hangs on the run's SETTINGS, one node whose backward
            # is one launch (cheetah_amd._chxtorch RunMomentEntry); the map is a constant of that node
            _, energy, distinct, meta, mass, nq, maps = run
            out, mom_y = _lib.torch_host().run_moment_entry(energy, distinct, lin.R.detach(), mom_x, partials, known, maps, meta, mass, nq,
                                                            index, take_sqrt, particles.shape[0])
        else:
            out, mom_y = _lib.torch_host().moment_entry_mapped(lin.R, particles.detach(), survival, mom_x, known, index, take_sqrt, partials)
        if known is None:
            particles._chx_mom = (particles._version, w_src, w_ver, particles.data_ptr(), mom_y)
        return out
    sshape = survival.shape[:-1] if survival is not None else ()
    batch_shape = bshapes(particles.shape[:-2], sshape)
    if tuple(batch_shape) != tuple(lin.batch_shape):
        return None
    B = numel(batch_shape)
    y, _ = flat_bcast(particles.detach(), batch_shape, 2)
    y = aligned(y)
    w = None
    if survival is not None:
        w, _ = flat_bcast(survival if survival.dtype == particles.dtype else survival.to(particles.dtype), batch_shape, 1)
        w = w.contiguous()
    if B == 1 and lin.R.shape[0] == 1 and lin.R.dtype == y.dtype and lin.x.shape[0] == 1:
        # one beam through one map: the node in C++ (cheetah_amd._chxtorch MomentEntryMappedNode: no Python frame in the forward
        # call's autograd bookkeeping and none at all in the backward pass)
        mom_x = _incoming_moments(lin, survival, w, B)
        w_src = None if survival is None else _origin(survival)
        cached = getattr(particles, "_chx_mom", None) if (not CAPTURING[0] or CAPTURE_KEEPS_BEAM_MOMENTS[0]) else None
        known = None
        if cached is not None and cached[0] == particles._version and cached[1] is w_src \
                and (w_src is None or cached[2] == w_src._version) and cached[3] == y.data_ptr() and cached[4].shape[0] == 1:
            known = cached[4]
        out, mom_y = _lib.torch_host().moment_entry_mapped(lin.R, y, w, mom_x, known, index, take_sqrt, None)
        if known is None:
            particles._chx_mom = (particles._version, w_src, None if w_src is None else w_src._version, y.data_ptr(), mom_y)
        return out.reshape(batch_shape)
    mom_y, picked = _memo_moments(particles, y, w, survival, B, (index, take_sqrt))
    if picked is not None and picked.dtype != lin.R.dtype:
        picked = None
    return MomentEntryMapped.apply(lin.R, mom_y, (lin, survival, w), index, take_sqrt, B, picked).reshape(batch_shape)


def moments(particles: torch.Tensor, survival: torch.Tensor | None, group=None) -> torch.Tensor:
    """(…,29) float64: [W, W2, mu(6), cov upper triangle (21)] for every vector entry. `group`: the particles are one rank's
    shard of a beam spread over that process group — the moments are those of ALL shards (a collective; see `Moments`)."""
    require_device(particles)
    N = particles.shape[-2]
    sshape = survival.shape[:-1] if survival is not None else ()
    batch_shape = bshapes(particles.shape[:-2], sshape)
    B = numel(batch_shape)
    x, _ = flat_bcast(particles, batch_shape, 2)
    x = aligned(x)
    w = None
    if survival is not None:
        w, _ = flat_bcast(survival.to(particles.dtype), batch_shape, 1)
        w = w.contiguous()
    lin = getattr(particles, "_chx_lin", None)
    if group is not None:
        # (the algebraic backward of a linearly tracked beam would hand every rank the WHOLE gradient of the map; the rows'
        # backward pass hands it its shard's share, like every other differentiable path of a sharded beam)
        if torch.is_grad_enabled() and (x.requires_grad or (w is not None and w.requires_grad)):
            out = Moments.apply(x, w, B, group)
        else:
            from . import sharding

            out = sharding.gather_merge_moments(_moments_raw(x, w, B, N), group)
    elif lin is not None and torch.is_grad_enabled() and lin.version == particles._version and x.requires_grad \
            and not (w is not None and w.requires_grad) and tuple(lin.batch_shape) == tuple(batch_shape) \
            and lin.R.requires_grad:
        # a moment of a linearly tracked beam whose particles carry no graph: the gradient reaches the map through
        # mu' = A mu + b, cov' = A C A^T (MomentsMapped) — no particle-sized backward pass
        out = MomentsMapped.apply(lin.R, x.detach(), w, (lin, survival), B)
    elif torch.is_grad_enabled() and (x.requires_grad or (w is not None and w.requires_grad)):
        out = Moments.apply(x, w, B)
    else:
        out = _moments_raw(x, w, B, N)
    return out.reshape(*batch_shape, MOM_NOUT)


def aperture_mask(particles: torch.Tensor, survival: torch.Tensor, x_max: torch.Tensor, y_max: torch.Tensor,
                  shape: str) -> torch.Tensor:
    """survival * inside(x, y) (aperture.py:104-128, chx_aperture_mask) with torch broadcasting of the vector dims."""
    require_device(particles, survival, x_max, y_max)
    if shape not in ("rectangular", "elliptical"):
        raise AssertionError(f"Unknown aperture shape {shape}")
    dt = particles.dtype
    N = particles.shape[-2]
    lim_shape = bshapes(x_max.shape, y_max.shape)
    batch_shape = bshapes(particles.shape[:-2], survival.shape[:-1], lim_shape)
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    s, Bs = flat_bcast(survival.to(dt), batch_shape, 1)
    limits = torch.stack(torch.broadcast_tensors(x_max.to(dt), y_max.to(dt)), dim=-1)
    lim, Bl = flat_bcast(limits, batch_shape, 1)
    x, s, lim = x.contiguous(), s.contiguous(), lim.contiguous()
    out = torch.empty((B, N), dtype=dt, device=x.device)
    check(_lib.lib().chx_aperture_mask(ptr(x), ptr(s), ptr(lim), 0 if shape == "rectangular" else 1, B, Bx, Bs, Bl, N,
                                       dtype_code(dt), ptr(out), stream_ptr()), "chx_aperture_mask")
    return out.reshape(*batch_shape, N)


def track_moments(particles: torch.Tensor, survival: torch.Tensor | None, tm: torch.Tensor) -> torch.Tensor:
    """Moments (…,29) of `particles @ tm.mT` without materialising the tracked particles (chx_track_moments)."""
    require_device(particles, tm)
    if particles.requires_grad or tm.requires_grad:
        return moments(apply_map(particles, tm), survival)  # differentiable path: track, then reduce
    if tm.dtype != particles.dtype:
        raise RuntimeError(f"transfer map dtype {tm.dtype} does not match particle dtype {particles.dtype}")
    N = particles.shape[-2]
    sshape = survival.shape[:-1] if survival is not None else ()
    in_shape = bshapes(particles.shape[:-2], sshape)
    batch_shape = bshapes(in_shape, tm.shape[:-2])
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    R, BR = flat_bcast(tm, batch_shape, 2)
    x, R = aligned(x), R.contiguous()
    w, Bw = None, 1
    if survival is not None:
        w, Bw = flat_bcast(survival.to(particles.dtype), batch_shape, 1)
        w = w.contiguous()
    # shift point of the one-pass second moments: the incoming means (one cheap pass over the input rows)
    sums = moment_sums(x, w if (Bw == 1 or Bw == Bx) else None, Bx)
    centre = (sums[:, 2:8] / sums[:, 0:1]).contiguous()
    lib = _lib.lib()
    ws_bytes = lib.chx_track_moments_workspace_bytes(B, N)
    ws = workspace(ws_bytes, x.device)
    out = torch.empty((B, MOM_NOUT), dtype=torch.float64, device=x.device)
    check(lib.chx_track_moments(ptr(x), ptr(w), ptr(R), ptr(centre), B, Bx, BR, Bw, N, dtype_code(x.dtype), ptr(out),
                                ptr(ws), ws_bytes, stream_ptr()), "chx_track_moments")
    return out.reshape(*batch_shape, MOM_NOUT)


# ---------------------------------------------------------------------------------------------
# ParameterBeam path
class ParameterTrack(torch.autograd.Function):
    """(mu', cov') = (R mu, R cov R^T) per batch row (chx_parameter_track without cavity coefficients) with a HIP backward
    (chx_parameter_track_bwd) — one node instead of the three matmul nodes of element.py:167-179."""

    @staticmethod
    def forward(ctx, mu, cov, tm, batch_shape):
        B = numel(batch_shape)
        m, Bm = flat_bcast(mu, batch_shape, 1)
        c, Bc = flat_bcast(cov, batch_shape, 2)
        R, BR = flat_bcast(tm, batch_shape, 2)
        m, c, R = m.contiguous(), c.contiguous(), R.contiguous()
        mu_out = torch.empty((B, 7), dtype=mu.dtype, device=mu.device)
        cov_out = torch.empty((B, 7, 7), dtype=mu.dtype, device=mu.device)
        check(_lib.lib().chx_parameter_track(ptr(m), ptr(c), ptr(R), None, B, Bm, Bc, BR, dtype_code(mu.dtype), ptr(mu_out),
                                             ptr(cov_out), stream_ptr()), "chx_parameter_track")
        ctx.save_for_backward(m, c, R)
        ctx.meta = (tuple(batch_shape), B, Bm, Bc, BR, mu.shape, cov.shape, tm.shape)
        mu_out, cov_out = mu_out.reshape(*batch_shape, 7), cov_out.reshape(*batch_shape, 7, 7)
        # the graph flags of the reference's two expressions: mu' hangs on (mu, tm), cov' on (cov, tm)
        need_mu, need_cov, need_tm = ctx.needs_input_grad[:3]
        dead = [t for t, alive in ((mu_out, need_mu or need_tm), (cov_out, need_cov or need_tm)) if not alive]
        if dead:
            ctx.mark_non_differentiable(*dead)
        return mu_out, cov_out

    @staticmethod
    def backward(ctx, g_mu, g_cov):
        m, c, R = ctx.saved_tensors
        batch_shape, B, Bm, Bc, BR, mu_shape, cov_shape, tm_shape = ctx.meta
        need = ctx.needs_input_grad
        dt, dev = m.dtype, m.device
        gm = g_mu.to(dt).reshape(B, 7).contiguous() if g_mu is not None else None
        gc = g_cov.to(dt).reshape(B, 7, 7).contiguous() if g_cov is not None else None
        d_mu = torch.empty((B, 7), dtype=dt, device=dev) if need[0] else None
        d_cov = torch.empty((B, 7, 7), dtype=dt, device=dev) if need[1] else None
        d_R = torch.empty((B, 7, 7), dtype=dt, device=dev) if need[2] else None
        check(_lib.lib().chx_parameter_track_bwd(ptr(gm), ptr(gc), ptr(m), ptr(c), ptr(R), B, Bm, Bc, BR, dtype_code(dt), ptr(d_mu),
                                                 ptr(d_cov), ptr(d_R), stream_ptr()), "chx_parameter_track_bwd")
        # one gradient row per batch row: fold the rows of a broadcast input
        if d_mu is not None:
            d_mu = d_mu.reshape(*batch_shape, 7).sum_to_size(mu_shape)
        if d_cov is not None:
            d_cov = d_cov.reshape(*batch_shape, 7, 7).sum_to_size(cov_shape)
        if d_R is not None:
            d_R = d_R.reshape(*batch_shape, 7, 7).sum_to_size(tm_shape)
        return d_mu, d_cov, d_R, None


def parameter_track(mu, cov, tm, cavity_coeffs=None, batch_shape=None):
    """mu (…,7), cov (…,7,7), tm (…,7,7) -> (mu', cov') = (tm mu, tm cov tm^T) (element.py:167-179)."""
    require_device(mu, cov, tm)
    if tm.dtype != mu.dtype:
        # element.py:170 `tm @ mu`: torch refuses a float32 map on float64 moments (and the other way round)
        raise RuntimeError(f"expected m1 and m2 to have the same dtype, but got: {tm.dtype} != {mu.dtype} "
                           "(transfer map vs beam moments)")
    if mu.requires_grad or cov.requires_grad or tm.requires_grad or (
            cavity_coeffs is not None and cavity_coeffs.requires_grad):
        # gradient path (tests/test_differentiable.py:58-75): one node with a HIP backward
        if batch_shape is None:
            batch_shape = bshapes(mu.shape[:-1], cov.shape[:-2], tm.shape[:-2])
        mu_out, cov_out = ParameterTrack.apply(mu, cov.to(mu.dtype), tm, tuple(batch_shape))
        if cavity_coeffs is None:
            return mu_out, cov_out
        # active cavity: the moment updates of parameter_track_kernel as (B,)-sized tensor expressions
        # (cavity.py:127-133, 202-218); cf = [a, b, k beta0, phi, cos phi, T566, T556, T555] in fp64
        cf = cavity_coeffs.reshape(*batch_shape, cavity_coeffs.shape[-1])
        dt = mu.dtype
        mu4, mu5 = mu[..., 4].double(), mu[..., 5].double()
        c44, c45, c55 = cov[..., 4, 4].double(), cov[..., 4, 5].double(), cov[..., 5, 5].double()
        new5 = mu5 * cf[..., 0] + cf[..., 1] * (torch.cos(-mu4 * cf[..., 2] + cf[..., 3]) - cf[..., 4])
        add4 = cf[..., 5] * mu5 * mu5 + cf[..., 6] * mu4 * mu5 + cf[..., 7] * mu4 * mu4
        q = (cf[..., 5] * c55 * c55 + cf[..., 6] * c45 * c55 + cf[..., 7] * c44 * c44).to(dt)
        mu_out = mu_out.expand(*batch_shape, 7)
        mu_out = torch.cat([mu_out[..., :4], (mu_out[..., 4].double() + add4).to(dt).unsqueeze(-1),
                            new5.to(dt).expand(batch_shape).unsqueeze(-1), mu_out[..., 6:]], dim=-1)
        cov_out = cov_out.expand(*batch_shape, 7, 7).clone()
        cov_out[..., 5, 5] = c55.to(dt)
        cov_out[..., 4, 4] = q
        cov_out[..., 4, 5] = q
        cov_out[..., 5, 4] = q
        return mu_out, cov_out
    if batch_shape is None:
        batch_shape = bshapes(mu.shape[:-1], cov.shape[:-2], tm.shape[:-2])
    B = numel(batch_shape)
    m, Bm = flat_bcast(mu, batch_shape, 1)
    c, Bc = flat_bcast(cov, batch_shape, 2)
    R, BR = flat_bcast(tm.to(mu.dtype), batch_shape, 2)
    m, c, R = m.contiguous(), c.contiguous(), R.contiguous()
    mu_out = torch.empty((B, 7), dtype=mu.dtype, device=mu.device)
    cov_out = torch.empty((B, 7, 7), dtype=mu.dtype, device=mu.device)
    check(_lib.lib().chx_parameter_track(ptr(m), ptr(c), ptr(R), ptr(cavity_coeffs), B, Bm, Bc, BR,
                                         dtype_code(mu.dtype), ptr(mu_out), ptr(cov_out), stream_ptr()),
          "chx_parameter_track")
    return mu_out.reshape(*batch_shape, 7), cov_out.reshape(*batch_shape, 7, 7)


def screen_gaussian(mu, cov, shift, geom, width: int, height: int) -> torch.Tensor:
    """Bivariate-normal screen image (…, height, width) of a ParameterBeam (screen.py:255-291)."""
    require_device(mu, cov, geom)
    batch_shape = bshapes(mu.shape[:-1], cov.shape[:-2], shift.shape[:-1] if shift is not None else ())
    B = numel(batch_shape)
    m, Bm = flat_bcast(mu, batch_shape, 1)
    c, Bc = flat_bcast(cov, batch_shape, 2)
    m, c = m.contiguous(), c.contiguous()
    sh, Bsh = (None, 1)
    if shift is not None:
        sh, Bsh = flat_bcast(shift.to(mu.dtype), batch_shape, 1)
        sh = sh.contiguous()
    geom = geom.to(mu.dtype).contiguous()
    img = torch.empty((B, height, width), dtype=mu.dtype, device=mu.device)
    # exact (working-precision) sample positions; the reference's dtype-less torch.arange grid carries fp32
    # jitter (see chx_parameter.hip), which only matters for beams far narrower than a pixel
    pos_f32 = 0
    check(_lib.lib().chx_screen_gaussian(ptr(m), ptr(c), ptr(sh), ptr(geom), B, Bm, Bc, Bsh, width, height, pos_f32,
                                         dtype_code(mu.dtype), ptr(img), stream_ptr()), "chx_screen_gaussian")
    return img.reshape(*batch_shape, height, width)


# ---------------------------------------------------------------------------------------------
# split moment passes (multi-GPU: all-reduce the partials between the passes)
def moment_sums(x, w, B):
    """Pass 1: (B,8) float64 [sum w, sum w^2, sum w x_0..5] of the LOCAL particles."""
    lib = _lib.lib()
    N = x.shape[1]
    ws_bytes = lib.chx_moments_workspace_bytes(B, N)
    ws = workspace(ws_bytes, x.device)
    sums = torch.empty((B, 8), dtype=torch.float64, device=x.device)
    check(lib.chx_moment_sums(ptr(x), ptr(w), B, x.shape[0], 1 if w is None else w.shape[0], N,
                              dtype_code(x.dtype), ptr(sums), ptr(ws), ws_bytes, stream_ptr()), "chx_moment_sums")
    return sums


def moment_centred(x, w, sums, B):
    """Pass 2: (B,21) float64 centred second-moment sums of the LOCAL particles about the GLOBAL mean
    encoded in `sums` (already all-reduced)."""
    lib = _lib.lib()
    N = x.shape[1]
    ws_bytes = lib.chx_moments_workspace_bytes(B, N)
    ws = workspace(ws_bytes, x.device)
    m2 = torch.empty((B, 21), dtype=torch.float64, device=x.device)
    check(lib.chx_moment_centred(ptr(x), ptr(w), ptr(sums), B, x.shape[0], 1 if w is None else w.shape[0], N,
                                 dtype_code(x.dtype), ptr(m2), ptr(ws), ws_bytes, stream_ptr()), "chx_moment_centred")
    return m2


def moment_finalize(sums, m2):
    B = sums.shape[0]
    out = torch.empty((B, MOM_NOUT), dtype=torch.float64, device=sums.device)
    check(_lib.lib().chx_moment_finalize(ptr(sums), ptr(m2), B, ptr(out), stream_ptr()), "chx_moment_finalize")
    return out


# ---------------------------------------------------------------------------------------------
# space charge (SpaceChargeKick's calls and autograd nodes) lives in _ops_sc.py; its names are part of this module's namespace
from ._ops_sc import *  # noqa: E402,F401,F403
from ._ops_sc import _bins3, _si  # noqa: E402,F401
# ... and the deposits (cloud in cell, histogram, kernel density) and special functions in _ops_deposit.py
from ._ops_deposit import *  # noqa: E402,F401,F403
from ._ops_deposit import _cic_args, _cic_deposit_raw, _hist_args, _launch_cic  # noqa: E402,F401

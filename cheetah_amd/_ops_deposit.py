"""Deposits and special functions of `cheetah_amd._ops`: the cloud-in-cell deposit in 1 / 2 / 3 dimensions and its autograd node
(utils/cloud_in_cell.py:8-451), `torch.histogramdd`'s 2-D histogram on explicit edges (screen.py:292-311), the kernel-density images
(utils/kde.py) and the reference's singularity-free special functions (utils/autograd.py:4-700): thin callers of `chx_cic_*`,
`chx_hist2d`, `chx_kde_*`, `chx_special`.

Part of `_ops` (which re-exports every name here: callers keep writing `_ops.cic_deposit(...)`); split out of that module in
round 6 for its size. Imported at the END of `_ops`, whose helpers it uses."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import CicArgs, Hist2dArgs
from ._ops import check, dtype_code, flat_bcast, numel, ptr, require_device, stream_ptr, workspace

__all__ = ['_cic_args', '_launch_cic', 'CicDeposit', 'cic_deposit', '_cic_deposit_raw', 'cic_deposit_mapped', 'compose_prefix', 'cic_deposit_into', 'sc_deposit_overwrite', 'cic_indices', '_hist_args', 'hist2d', 'hist2d_indices', 'special', 'kde_histogram_2d', 'kde_histogram_1d', 'SORTED_CIC_MIN_PARTICLES', 'SPECIAL_KINDS', 'KDE_CHUNK']


def _cic_args(particles, cols, bins, extent, charge, survival, scale, shift, abs_charge, grid=None,
              grid_strides=None, grid_batch_stride=0, extra_batch=()):
    require_device(particles, extent)
    dt = particles.dtype
    N = particles.shape[-2]
    nd = len(cols)
    keep = []
    # no vector dimension anywhere (one beam on one screen / grid: the control-loop case): nothing to broadcast
    plain = (particles.dim() == 2 and extent.dim() == 2 and not extra_batch
             and all(t is None or t.dim() == 1 for t in (charge, survival, scale, shift)))
    if plain:
        batch_shape, B = (), 1

        def prep(t, n_tail):
            if t is None:
                return None, 1
            f = t if t.dtype == dt else t.to(dt)
            f = f if f.is_contiguous() else f.contiguous()
            keep.append(f)
            return f, 1
    else:
        shapes = [particles.shape[:-2], extent.shape[:-2], tuple(extra_batch)]
        for t, k in ((charge, 1), (survival, 1), (scale, 1), (shift, 1)):
            if t is not None:
                shapes.append(t.shape[:-1])
        batch_shape = torch.broadcast_shapes(*shapes)
        B = numel(batch_shape)

        def prep(t, n_tail):
            if t is None:
                return None, 1
            f, Bt = flat_bcast(t.to(dt), batch_shape, n_tail)
            f = f.contiguous()
            keep.append(f)
            return f, Bt

    x, Bx = prep(particles, 2)
    ext, Be = prep(extent, 2)
    q, Bq = prep(charge, 1)
    s, Bs = prep(survival, 1)
    sc, Bsc = prep(scale, 1)
    sh, Bsh = prep(shift, 1)
    a = CicArgs()
    a.ndim = nd
    for d in range(nd):
        a.cols[d] = int(cols[d])
        a.bins[d] = int(bins[d])
        a.grid_strides[d] = 0 if grid_strides is None else int(grid_strides[d])
    a.grid_batch_stride = int(grid_batch_stride)
    a.B, a.Bx, a.Bq, a.Bs, a.Be, a.Bsc, a.Bsh, a.N = B, Bx, Bq, Bs, Be, Bsc, Bsh, N
    a.dtype = dtype_code(dt)
    a.abs_charge = int(bool(abs_charge))
    a.x, a.charge, a.survival, a.extent = ptr(x), ptr(q), ptr(s), ptr(ext)
    a.scale, a.shift = ptr(sc), ptr(sh)
    a.grid = ptr(grid)
    return a, keep, batch_shape, B, N


SORTED_CIC_MIN_PARTICLES = 65536  # below this the 2^d global atomics per particle are cheaper than 5 launches


def _launch_cic(a: CicArgs, N: int, ndim: int, device, mode: str = "auto") -> None:
    """Direct (global float atomics) or sorted / LDS-privatised deposit (chx_cic_deposit_sorted)."""
    lib = _lib.lib()
    use_sorted = mode == "sorted" or (mode == "auto" and ndim >= 2 and N >= SORTED_CIC_MIN_PARTICLES)
    if use_sorted:
        nbytes = lib.chx_cic_sorted_workspace_bytes(ctypes.byref(a))
        if nbytes > 0:
            ws = workspace(nbytes, device)
            st = lib.chx_cic_deposit_sorted(ctypes.byref(a), ptr(ws), nbytes, stream_ptr())
            if st == 0:
                return
            if mode == "sorted":
                check(st, "chx_cic_deposit_sorted")
        elif mode == "sorted":
            raise _lib.ChxError("sorted deposit needs ndim >= 2")
    check(lib.chx_cic_deposit(ctypes.byref(a), stream_ptr()), "chx_cic_deposit")


class CicDeposit(torch.autograd.Function):
    """Differentiable deposit (the reference's CIC is differentiable through torch ops,
    utils/cloud_in_cell.py:14); backward = chx_cic_deposit_bwd: gradient wrt the particle coordinates
    (through the corner weights) and wrt charges / survival probabilities."""

    @staticmethod
    def forward(ctx, particles, charge, survival, cols, bins, extent, scale, shift, abs_charge, transpose_2d, mode):
        ctx.save_for_backward(particles, charge, survival, extent, scale, shift)
        ctx.meta = (tuple(cols), tuple(bins), abs_charge, transpose_2d)
        return _cic_deposit_raw(particles, cols, bins, extent, charge, survival, scale, shift, abs_charge,
                                transpose_2d, mode)

    @staticmethod
    def backward(ctx, dgrid):
        particles, charge, survival, extent, scale, shift = ctx.saved_tensors
        cols, bins, abs_charge, transpose_2d = ctx.meta
        nd = len(cols)
        a, keep, batch_shape, B, N = _cic_args(particles.detach(), cols, bins, extent, charge, survival, scale, shift,
                                               abs_charge)
        total = numel(bins)
        if transpose_2d:
            a.grid_strides[0], a.grid_strides[1] = 1, bins[0]
        a.grid_batch_stride = total
        dg = dgrid.to(particles.dtype).reshape(B, total).contiguous()
        dweight = torch.empty((B, N), dtype=particles.dtype, device=particles.device)
        dpos = torch.empty((B, N, nd), dtype=particles.dtype, device=particles.device)
        check(_lib.lib().chx_cic_deposit_bwd(ctypes.byref(a), ptr(dg), ptr(dweight), ptr(dpos), stream_ptr()),
              "chx_cic_deposit_bwd")
        dweight = dweight.reshape(*batch_shape, N)
        dpos = dpos.reshape(*batch_shape, N, nd)
        d_particles = d_charge = d_survival = d_extent = d_scale = None
        if ctx.needs_input_grad[5] or ctx.needs_input_grad[6]:
            # bin-space position pb = (v - l) / (r - l) * bins - 0.5 with v = scale * x - shift, and dpos = dL/dv:
            #   dL/dl = -sum dpos (r - v) / (r - l),  dL/dr = -sum dpos (v - l) / (r - l),  dL/dscale = sum dpos x
            xc = particles.detach()[..., list(cols)]
            v = xc * scale.unsqueeze(-2) if scale is not None else xc
            if shift is not None:
                v = v - shift.unsqueeze(-2)
            if ctx.needs_input_grad[5]:
                lo, hi = extent[..., 0].unsqueeze(-2), extent[..., 1].unsqueeze(-2)
                d_lo = -(dpos * (hi - v) / (hi - lo)).sum(dim=-2)
                d_hi = -(dpos * (v - lo) / (hi - lo)).sum(dim=-2)
                d_extent = torch.stack([d_lo, d_hi], dim=-1).sum_to_size(extent.shape)
            if scale is not None and ctx.needs_input_grad[6]:
                d_scale = (dpos * xc).sum(dim=-2).sum_to_size(scale.shape)
        if ctx.needs_input_grad[0]:
            if scale is not None:
                dpos = dpos * scale.unsqueeze(-2)
            full = torch.zeros((*batch_shape, N, 7), dtype=particles.dtype, device=particles.device)
            for d, c in enumerate(cols):
                full[..., c] = dpos[..., d]
            d_particles = full.sum_to_size(particles.shape)
        if charge is not None and ctx.needs_input_grad[1]:
            s_ = survival if survival is not None else 1.0
            d_charge = (dweight * s_ * (charge.sign() if abs_charge else 1.0)).sum_to_size(charge.shape)
        if survival is not None and ctx.needs_input_grad[2]:
            c_ = (charge.abs() if abs_charge else charge) if charge is not None else 1.0
            d_survival = (dweight * c_).sum_to_size(survival.shape)
        return d_particles, d_charge, d_survival, None, None, d_extent, d_scale, None, None, None, None


def cic_deposit(particles, cols, bins, extent, charge=None, survival=None, scale=None, shift=None,
                abs_charge=False, transpose_2d=False, mode: str = "auto") -> torch.Tensor:
    """Cloud-in-cell deposition (utils/cloud_in_cell.py:8-451) of columns `cols` of the 7-vectors.

    Returns (*batch, *bins); with transpose_2d the 2-D image is written directly as (bins[1], bins[0])
    (the `.mT` of screen.py:339). Differentiable wrt particles / charge / survival.
    """
    if particles.requires_grad or (charge is not None and charge.requires_grad) or (
            survival is not None and survival.requires_grad) or extent.requires_grad or (
            scale is not None and scale.requires_grad):
        return CicDeposit.apply(particles, charge, survival, cols, bins, extent, scale, shift, abs_charge,
                                transpose_2d, mode)
    return _cic_deposit_raw(particles, cols, bins, extent, charge, survival, scale, shift, abs_charge, transpose_2d,
                            mode)


def _cic_deposit_raw(particles, cols, bins, extent, charge, survival, scale, shift, abs_charge, transpose_2d,
                     mode) -> torch.Tensor:
    bins = [int(b) for b in bins]
    a, keep, batch_shape, B, N = _cic_args(particles, cols, bins, extent, charge, survival, scale, shift, abs_charge)
    total = numel(bins)
    grid = torch.zeros((B, total), dtype=particles.dtype, device=particles.device)
    a.grid = ptr(grid)
    if transpose_2d:
        assert len(bins) == 2
        a.grid_strides[0], a.grid_strides[1] = 1, bins[0]
        a.grid_batch_stride = total
        out_shape = (bins[1], bins[0])
    else:
        out_shape = tuple(bins)
    _launch_cic(a, N, len(bins), particles.device, mode)
    return grid.reshape(*batch_shape, *out_shape)


def cic_deposit_mapped(particles, tm, cols, bins, extent, charge=None, survival=None, shift=None, abs_charge=False,
                       transpose_2d=False) -> torch.Tensor:
    """Deposit of the TRACKED beam `particles @ tm.mT` without forming it (chx_cic_deposit_mapped): the (…,N,7) output of a
    scan of lattice settings is never written, one image per setting comes back. Same cell indices and addends as
    `cic_deposit(apply_map(particles, tm), …)`."""
    require_device(particles, tm)
    if tm.dtype != particles.dtype:
        raise RuntimeError(f"transfer map dtype {tm.dtype} does not match particle dtype {particles.dtype}")
    bins = [int(b) for b in bins]
    a, keep, batch_shape, B, N = _cic_args(particles, cols, bins, extent, charge, survival, None, shift, abs_charge,
                                           extra_batch=tm.shape[:-2])
    R, BR = flat_bcast(tm, batch_shape, 2)
    R = R.contiguous()
    total = numel(bins)
    grid = torch.zeros((B, total), dtype=particles.dtype, device=particles.device)
    a.grid = ptr(grid)
    if transpose_2d:
        assert len(bins) == 2
        a.grid_strides[0], a.grid_strides[1] = 1, bins[0]
        a.grid_batch_stride = total
        out_shape = (bins[1], bins[0])
    else:
        out_shape = tuple(bins)
    check(_lib.lib().chx_cic_deposit_mapped(ctypes.byref(a), ptr(R), BR, stream_ptr()), "chx_cic_deposit_mapped")
    return grid.reshape(*batch_shape, *out_shape)


def compose_prefix(stack: torch.Tensor) -> torch.Tensor:
    """(E, Bm, 7, 7) per-element maps -> (E, Bm, 7, 7) prefix products M_e ... M_0 (chx_compose_prefix)."""
    E, Bm = stack.shape[0], stack.shape[1]
    stack = stack.contiguous()
    out = torch.empty_like(stack)
    check(_lib.lib().chx_compose_prefix(ptr(stack), E, Bm, Bm, dtype_code(stack.dtype), ptr(out), stream_ptr()),
          "chx_compose_prefix")
    return out


def cic_deposit_into(grid: torch.Tensor, grid_strides, grid_batch_stride, particles, cols, bins, extent,
                     charge=None, survival=None, scale=None, shift=None, mode: str = "auto") -> None:
    """Deposit into a caller-provided (zeroed) strided grid, e.g. the doubled Hockney array."""
    a, keep, batch_shape, B, N = _cic_args(particles, cols, bins, extent, charge, survival, scale, shift, False,
                                           grid=grid, grid_strides=grid_strides,
                                           grid_batch_stride=grid_batch_stride)
    _launch_cic(a, N, len(bins), particles.device, mode)


def sc_deposit_overwrite(x, q, w, extent, scale, B: int, N: int, bins) -> torch.Tensor:
    """The charge grid (B, gx, gy, gz) of a kick from x (Bx,N,7), charges, survival weights and the geometry kernel's
    extent (B,3,2) / scale (B,3), as `chx_sc_kick` deposits it: `chx_cic_deposit_sorted_overwrite` (every cell stored by the
    tile that owns it, no zero-fill) from 65536 particles on, below that a zeroed grid + the direct deposit."""
    g = [int(b) for b in bins]
    total = g[0] * g[1] * g[2]
    rho = torch.empty((B, *g), dtype=x.dtype, device=x.device)
    a, keep, _, _, _ = _cic_args(x, (0, 2, 4), g, extent, q, w, scale, None, False, grid=rho,
                                 grid_strides=(g[1] * g[2], g[2], 1), grid_batch_stride=total)
    lib = _lib.lib()
    if N >= SORTED_CIC_MIN_PARTICLES:
        nbytes = lib.chx_cic_sorted_workspace_bytes(ctypes.byref(a))
        ws = workspace(nbytes, x.device)
        check(lib.chx_cic_deposit_sorted_overwrite(ctypes.byref(a), ptr(ws), nbytes, stream_ptr()),
              "chx_cic_deposit_sorted_overwrite")
    else:
        rho.zero_()
        check(lib.chx_cic_deposit(ctypes.byref(a), stream_ptr()), "chx_cic_deposit")
    return rho


def cic_indices(particles, cols, bins, extent, scale=None, shift=None):
    a, keep, batch_shape, B, N = _cic_args(particles, cols, bins, extent, None, None, scale, shift, False)
    idx = torch.empty((B, N, len(cols)), dtype=torch.int32, device=particles.device)
    frac = torch.empty((B, N, len(cols)), dtype=particles.dtype, device=particles.device)
    check(_lib.lib().chx_cic_indices(ctypes.byref(a), ptr(idx), ptr(frac), stream_ptr()), "chx_cic_indices")
    return idx.reshape(*batch_shape, N, len(cols)), frac.reshape(*batch_shape, N, len(cols))


def _hist_args(particles, edges_x, edges_y, charge, survival, shift):
    require_device(particles, edges_x, edges_y)
    dt = particles.dtype
    N = particles.shape[-2]
    shapes = [particles.shape[:-2]]
    for t in (charge, survival, shift):
        if t is not None:
            shapes.append(t.shape[:-1])
    batch_shape = torch.broadcast_shapes(*shapes)
    B = numel(batch_shape)
    keep = []

    def prep(t, n_tail):
        if t is None:
            return None, 1
        f, Bt = flat_bcast(t.to(dt), batch_shape, n_tail)
        f = f.contiguous()
        keep.append(f)
        return f, Bt

    x, Bx = prep(particles, 2)
    q, Bq = prep(charge, 1)
    s, Bs = prep(survival, 1)
    sh, Bsh = prep(shift, 1)
    ex, ey = edges_x.to(dt).contiguous(), edges_y.to(dt).contiguous()
    keep += [ex, ey]
    a = Hist2dArgs()
    a.B, a.Bx, a.Bq, a.Bs, a.Bsh, a.N = B, Bx, Bq, Bs, Bsh, N
    a.nx, a.ny = ex.shape[0] - 1, ey.shape[0] - 1
    a.dtype = dtype_code(dt)
    a.x, a.charge, a.survival, a.shift = ptr(x), ptr(q), ptr(s), ptr(sh)
    a.edges_x, a.edges_y = ptr(ex), ptr(ey)
    return a, keep, batch_shape, B, N


def hist2d(particles, edges_x, edges_y, charge=None, survival=None, shift=None) -> torch.Tensor:
    """Screen "histogram" image (…, ny, nx) (screen.py:292-311)."""
    a, keep, batch_shape, B, N = _hist_args(particles, edges_x, edges_y, charge, survival, shift)
    img = torch.zeros((B, a.ny, a.nx), dtype=particles.dtype, device=particles.device)
    a.image = ptr(img)
    check(_lib.lib().chx_hist2d(ctypes.byref(a), stream_ptr()), "chx_hist2d")
    return img.reshape(*batch_shape, a.ny, a.nx)


def hist2d_indices(particles, edges_x, edges_y, shift=None) -> torch.Tensor:
    a, keep, batch_shape, B, N = _hist_args(particles, edges_x, edges_y, None, None, shift)
    ij = torch.empty((B, N, 2), dtype=torch.int32, device=particles.device)
    check(_lib.lib().chx_hist2d_indices(ctypes.byref(a), ptr(ij), stream_ptr()), "chx_hist2d_indices")
    return ij.reshape(*batch_shape, N, 2)


# ---------------------------------------------------------------------------------------------
# space charge + SI conversions
SPECIAL_KINDS = {"log1pdiv": 0, "si1mdiv": 1, "sicos1mdiv": 2, "sipsicos3mdiv": 3, "sicoskuddelmuddel15mdiv": 4,
                 "cossqrtmcosdivdiff": 5, "simsidivdiff": 6, "si2msi2divdiff": 7, "sqrta2minusbdiva": 8}


def special(kind: str, a: torch.Tensor, b: torch.Tensor | None = None):
    """(f, df/da[, df/db]) of one of the reference's special functions (utils/autograd.py:4-74) element-wise, chx_special."""
    code = SPECIAL_KINDS[kind]
    if b is not None:
        a, b = torch.broadcast_tensors(a, b)
        require_device(a, b)
        b = b.contiguous()
    else:
        require_device(a)
    a = a.contiguous()
    out, da = torch.empty_like(a), torch.empty_like(a)
    db = torch.empty_like(a) if b is not None else None
    check(_lib.lib().chx_special(code, ptr(a), ptr(b), a.numel(), dtype_code(a.dtype), ptr(out), ptr(da), ptr(db),
                                 stream_ptr()), "chx_special")
    return (out, da) if b is None else (out, da, db)


KDE_CHUNK = 131072  # particles per GEMM slab: bounds the (N, bins) kernel-value arrays at ~1 GiB each


def kde_histogram_2d(particles, centres_x, centres_y, bandwidth, charge=None, survival=None, shift=None,
                     epsilon: float = 1e-10, group=None) -> torch.Tensor:
    """Screen "kde" image (…, H, W) (utils/kde.py:137-204 + the `.mT` of screen.py:326): chx_kde_values for the two
    sets of Gaussian kernel values, their GEMM over the particle axis (rocBLAS through torch.matmul), normalised to a
    pdf. With gradient tracking the kernel values are tensor expressions instead, so autograd sees them. `group`: the particles
    are one rank's shard of a beam (`sharding.particle_sharded`): the kernel sums are added over the ranks before the normalisation."""
    require_device(particles, centres_x, centres_y, bandwidth)
    dt = particles.dtype
    N = particles.shape[-2]
    shapes = [particles.shape[:-2]] + [t.shape[:-1] for t in (charge, survival, shift) if t is not None]
    batch_shape = torch.broadcast_shapes(*shapes)
    B = numel(batch_shape)
    flat = lambda t, k: (None, 1) if t is None else flat_bcast(t.to(dt), batch_shape, k)  # noqa: E731
    x, Bx = flat_bcast(particles, batch_shape, 2)
    q, Bq = flat(charge, 1)
    w, Bs = flat(survival, 1)
    sh, Bsh = flat(shift, 1)
    x = x.contiguous()
    q, w, sh = (None if t is None else t.contiguous() for t in (q, w, sh))
    cx, cy, sg = centres_x.to(dt).contiguous(), centres_y.to(dt).contiguous(), bandwidth.to(dt).reshape(1).contiguous()
    W_, H_ = cx.shape[0], cy.shape[0]
    differentiable = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, q, w, sh, sg))
    joint = torch.zeros((B, W_, H_), dtype=dt, device=x.device)
    tiny = torch.finfo(dt).tiny
    for n0 in range(0, N, KDE_CHUNK):
        nc = min(KDE_CHUNK, N - n0)
        if differentiable:
            def values(col, centres, weighted):
                v = x[:, n0:n0 + nc, col].expand(B, nc)
                if sh is not None:
                    v = v - sh[:, 0 if col == 0 else 1].reshape(-1, 1)
                k = (-0.5 * ((v.unsqueeze(-1) - centres) / sg).square()).exp() / (2 * torch.pi * sg.square()).sqrt()
                if weighted:
                    wt = torch.ones((), dtype=dt, device=x.device)
                    if q is not None:
                        wt = wt * q[:, n0:n0 + nc].abs()
                    if w is not None:
                        wt = wt * w[:, n0:n0 + nc]
                    k = wt.expand(B, nc).unsqueeze(-1) * k
                return k.clamp_min(tiny)
            k1, k2 = values(0, cx, True), values(2, cy, False)
        else:
            k1 = torch.empty((B, nc, W_), dtype=dt, device=x.device)
            k2 = torch.empty((B, nc, H_), dtype=dt, device=x.device)
            lib = _lib.lib()
            check(lib.chx_kde_values(ptr(x), ptr(q), ptr(w), ptr(sh), ptr(cx), ptr(sg), 0, B, Bx, Bq, Bs, Bsh, N, n0, nc, W_,
                                     dtype_code(dt), ptr(k1), stream_ptr()), "chx_kde_values")
            check(lib.chx_kde_values(ptr(x), None, None, ptr(sh), ptr(cy), ptr(sg), 2, B, Bx, 1, 1, Bsh, N, n0, nc, H_,
                                     dtype_code(dt), ptr(k2), stream_ptr()), "chx_kde_values")
        joint = joint + k1.mT @ k2
    if group is not None:
        # a particle-sharded beam: the kernel sums of ALL shards, then the normalisation (the pdf of the union, not of a shard)
        from . import sharding

        joint = sharding.sum_over_ranks(joint, group)
    pdf = joint / (joint.sum(dim=(-2, -1), keepdim=True) + epsilon)
    return pdf.mT.reshape(*batch_shape, H_, W_)


def kde_histogram_1d(particles, centres, bandwidth, survival=None, epsilon: float = 1e-10) -> torch.Tensor:
    """Normalised 1-D kernel density (…, len(centres)) of column 0 of `particles` (utils/kde.py:6-76,116-152):
    chx_kde_values for the Gaussian kernel values, summed over the particle axis slab by slab."""
    require_device(particles, centres, bandwidth)
    dt = particles.dtype
    N = particles.shape[-2]
    batch_shape = torch.broadcast_shapes(particles.shape[:-2], *([survival.shape[:-1]] if survival is not None else []))
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    x = x.contiguous()
    w, Bs = (None, 1) if survival is None else flat_bcast(survival.to(dt), batch_shape, 1)
    w = None if w is None else w.contiguous()
    c, sg = centres.to(dt).contiguous(), bandwidth.to(dt).reshape(1).contiguous()
    mass = torch.zeros((B, c.shape[0]), dtype=dt, device=x.device)
    for n0 in range(0, N, KDE_CHUNK):
        nc = min(KDE_CHUNK, N - n0)
        k = torch.empty((B, nc, c.shape[0]), dtype=dt, device=x.device)
        check(_lib.lib().chx_kde_values(ptr(x), None, ptr(w), None, ptr(c), ptr(sg), 0, B, Bx, 1, Bs, 1, N, n0, nc, c.shape[0],
                                        dtype_code(dt), ptr(k), stream_ptr()), "chx_kde_values")
        mass = mass + k.sum(dim=1)
    return (mass / (mass.sum(dim=-1, keepdim=True) + epsilon)).reshape(*batch_shape, c.shape[0])

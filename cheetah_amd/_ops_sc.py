"""Space-charge operations of `cheetah_amd._ops` (SpaceChargeKick: space_charge_kick.py:477-586 — grid geometry, integrated Green
function, pruned / dense Poisson solve, gather + kick, the tile-ordered chain) and the SI <-> Cheetah coordinate conversions
(particle_beam.py:1262-1346): thin callers of the `chx_sc_*` entry points and the autograd nodes of the differentiable kick.

Part of `_ops` (which re-exports every name here: callers keep writing `_ops.sc_kick(...)`); split out of that module in round 6
for its size. Imported at the END of `_ops`, whose helpers it uses."""
from __future__ import annotations

import ctypes

import torch

from . import _lib, _ops
from ._ops import MOM_NOUT, aligned, check, dtype_code, flat_bcast, numel, ptr, require_device, stream_ptr, workspace

__all__ = ['_bins3', 'sc_geometry', 'sc_igf', 'sc_pruned_supported', 'sc_green_spectrum', 'sc_convolve', 'sc_convolve_halo', 'ScFftPlan', 'sc_spectral_mul', 'sc_gradient', 'sc_kick', 'sc_tile_state', 'sc_kick_sorted', 'sc_gather_kick', 'sc_gather_kick_phi', 'ScPoisson', 'ScPoissonDense', 'ScGradient', 'ScGatherKick', '_si', 'to_xyz_pxpypz', 'from_xyz_pxpypz']


def _bins3(bins):
    return (ctypes.c_int32 * 3)(*[int(b) for b in bins])


def sc_geometry(mom, grid_extent, energy, length, mass_eV: float, pot_factor: float, B: int, bins):
    """One launch for the grid geometry of a kick (chx_sc_geometry): returns half (B,3), cell (B,3), gamma (B,),
    dt (B,), scale (B,3), extent (B,3,2) in the beam dtype and pot_scale (B,) float64."""
    dt_, dev = energy.dtype, energy.device
    buf = torch.empty(17 * B, dtype=dt_, device=dev)
    half, cell = buf[: 3 * B].view(B, 3), buf[3 * B: 6 * B].view(B, 3)
    gamma, dt = buf[6 * B: 7 * B], buf[7 * B: 8 * B]
    scale, extent = buf[8 * B: 11 * B].view(B, 3), buf[11 * B: 17 * B].view(B, 3, 2)
    pot_scale = torch.empty(B, dtype=torch.float64, device=dev)
    check(_lib.lib().chx_sc_geometry(ptr(mom), ptr(grid_extent), ptr(energy), ptr(length), mass_eV, pot_factor, B,
                                     mom.shape[0], grid_extent.shape[0], energy.shape[0], length.shape[0], _bins3(bins),
                                     dtype_code(dt_), ptr(half), ptr(cell), ptr(gamma), ptr(dt), ptr(scale), ptr(extent),
                                     ptr(pot_scale), stream_ptr()), "chx_sc_geometry")
    return half, cell, gamma, dt, scale, extent, pot_scale


def sc_igf(cell, gamma, bins, padded: bool = False) -> torch.Tensor:
    """Integrated Green function on the doubled grid, (B,2gx,2gy,2gz); `padded=True` returns the in-place R2C layout
    (B,2gx,2gy,2gz+2) (the two extra values per row are don't-care)."""
    B = cell.shape[0]
    lib = _lib.lib()
    b3 = _bins3(bins)
    ws_bytes = lib.chx_sc_igf_workspace_bytes(B, b3)
    ws = workspace(ws_bytes, cell.device)
    ldz = 2 * bins[2] + (2 if padded else 0)
    G = torch.empty((B, 2 * bins[0], 2 * bins[1], ldz), dtype=cell.dtype, device=cell.device)
    check(lib.chx_sc_igf(ptr(cell), ptr(gamma), B, b3, dtype_code(cell.dtype), ptr(G), ldz, ptr(ws), ws_bytes,
                         stream_ptr()), "chx_sc_igf")
    return G


def sc_pruned_supported(bins, dtype) -> bool:
    return bool(_lib.lib().chx_sc_pruned_supported(_bins3(bins), dtype_code(dtype)))


def sc_green_spectrum(cell, gamma, bins, exact: bool = False) -> torch.Tensor:
    """Real, even spectrum of the integrated Green function, (B, gx+1, gy+1, gz+1). Default: chx_sc_green_spectrum_fast
    (fp32: far cells by the multipole expansion of the cell integral, see include/chx.h); `exact=True`: corner table
    everywhere (chx_sc_igf_table + chx_sc_green_spectrum)."""
    B = cell.shape[0]
    lib = _lib.lib()
    b3 = _bins3(bins)
    dt = dtype_code(cell.dtype)
    n1 = (bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1)
    if not exact:
        ws_bytes = lib.chx_sc_green_fast_workspace_bytes(B, b3, dt)
        ws = workspace(ws_bytes, cell.device)
        Ghat = torch.empty((B, bins[0] + 1, bins[1] + 1, bins[2] + 1), dtype=cell.dtype, device=cell.device)
        check(lib.chx_sc_green_spectrum_fast(ptr(cell), ptr(gamma), B, b3, dt, ptr(Ghat), ptr(ws), ws_bytes, stream_ptr()),
              "chx_sc_green_spectrum_fast")
        return Ghat
    table = torch.empty((B, n1), dtype=torch.float64, device=cell.device)
    check(lib.chx_sc_igf_table(ptr(cell), ptr(gamma), B, b3, dt, ptr(table), stream_ptr()), "chx_sc_igf_table")
    ws_bytes = lib.chx_sc_green_workspace_bytes(B, b3, dt)
    ws = workspace(ws_bytes, cell.device)
    Ghat = torch.empty((B, bins[0] + 1, bins[1] + 1, bins[2] + 1), dtype=cell.dtype, device=cell.device)
    check(lib.chx_sc_green_spectrum(ptr(table), B, b3, dt, ptr(Ghat), ptr(ws), ws_bytes, stream_ptr()),
          "chx_sc_green_spectrum")
    return Ghat


def sc_convolve(rho, Ghat, scale, bins) -> torch.Tensor:
    """phi (B,gx,gy,gz) from the compact charge grid rho (B,gx,gy,gz) (chx_sc_convolve)."""
    B = rho.shape[0]
    lib = _lib.lib()
    b3 = _bins3(bins)
    dt = dtype_code(rho.dtype)
    ws_bytes = lib.chx_sc_convolve_workspace_bytes(B, b3, dt)
    ws = workspace(ws_bytes, rho.device)
    phi = torch.empty_like(rho)
    check(lib.chx_sc_convolve(ptr(rho), ptr(Ghat), ptr(scale), B, b3, dt, ptr(phi), ptr(ws), ws_bytes, stream_ptr()),
          "chx_sc_convolve")
    return phi


def sc_convolve_halo(rho, Ghat, scale, bins) -> torch.Tensor:
    """The potential of sc_convolve stored inside a halo of 2 nodes, (B,gx+4,gy+4,gz+4); the halo is not written
    (chx_sc_convolve_halo). Input of sc_gather_kick_phi."""
    B = rho.shape[0]
    lib = _lib.lib()
    b3 = _bins3(bins)
    dt = dtype_code(rho.dtype)
    ws_bytes = lib.chx_sc_convolve_workspace_bytes(B, b3, dt)
    ws = workspace(ws_bytes, rho.device)
    phi = torch.empty((B, bins[0] + 4, bins[1] + 4, bins[2] + 4), dtype=rho.dtype, device=rho.device)
    assert phi.numel() == lib.chx_sc_phi_halo_elements(B, b3)
    check(lib.chx_sc_convolve_halo(ptr(rho), ptr(Ghat), ptr(scale), B, b3, dt, ptr(phi), ptr(ws), ws_bytes, stream_ptr()),
          "chx_sc_convolve_halo")
    return phi


class ScFftPlan:
    """hipFFT plans of the Hockney convolution for one (B, grid, dtype) (chx_sc_fft_plan_*): in-place, unnormalised
    real <-> complex 3-D transforms on the padded layout (B,2gx,2gy,2gz+2)."""

    def __init__(self, B: int, bins, dtype: torch.dtype):
        handle = ctypes.c_void_p()
        check(_lib.lib().chx_sc_fft_plan_create(B, _bins3(bins), dtype_code(dtype), ctypes.byref(handle)),
              "chx_sc_fft_plan_create")
        self._handle = handle
        self.key = (B, tuple(bins), dtype)

    def forward(self, data: torch.Tensor, which: int = 0) -> None:
        check(_lib.lib().chx_sc_fft_exec(self._handle, which, ptr(data), stream_ptr()), "chx_sc_fft_exec")

    def inverse(self, data: torch.Tensor) -> None:
        check(_lib.lib().chx_sc_fft_exec(self._handle, 2, ptr(data), stream_ptr()), "chx_sc_fft_exec")

    def __del__(self):
        try:
            if self._handle:
                _lib.lib().chx_sc_fft_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def sc_spectral_mul(rho_hat, G_hat, scale) -> None:
    """rho_hat *= G_hat * scale; complex tensors, or real tensors holding interleaved (re, im) pairs in the last dim."""
    B = rho_hat.shape[0]
    n = numel(rho_hat.shape[1:])
    if rho_hat.is_complex():
        real_dtype = torch.float32 if rho_hat.dtype == torch.complex64 else torch.float64
    else:
        real_dtype, n = rho_hat.dtype, n // 2
    check(_lib.lib().chx_sc_spectral_mul(ptr(rho_hat), ptr(G_hat), ptr(scale), B, n, dtype_code(real_dtype),
                                         stream_ptr()), "chx_sc_spectral_mul")


def sc_gradient(phi, cell, gamma, bins) -> torch.Tensor:
    """phi: doubled (B,2gx,2gy,2gz) or compact (B,gx,gy,gz) potential -> packed force grid (B,gx,gy,gz,4)."""
    B = phi.shape[0]
    doubled = int(phi.shape[1] == 2 * bins[0])
    ldz = phi.shape[3] if doubled else 0  # 2gz, or 2gz + 2 after an in-place inverse transform
    F = torch.empty((B, bins[0], bins[1], bins[2], 4), dtype=phi.dtype, device=phi.device)
    check(_lib.lib().chx_sc_gradient(ptr(phi), ptr(cell), ptr(gamma), B, _bins3(bins), doubled, ldz,
                                     dtype_code(phi.dtype), ptr(F), stream_ptr()), "chx_sc_gradient")
    return F


def sc_kick(x, q, w, energy, length, grid_extent, mass_eV, B, N, bins, side_stream=None, post_map_ptr=None) -> torch.Tensor:
    """One chx_sc_kick call: x (Bx,N,7), q (Bq,N), w (Bs,N), energy (B,), length (B,), grid_extent (Bext,3) -> (B,N,7).
    `post_map_ptr`: device address of a (7,7) map of the beam dtype applied to the kicked particles in the same pass (the
    linear run that follows the kick in a Segment)."""
    lib = _lib.lib()
    b3 = _bins3(bins)
    dt = dtype_code(x.dtype)
    ws_bytes = lib.chx_sc_kick_workspace_bytes(B, N, b3, dt)
    ws = workspace(ws_bytes, x.device)
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(lib.chx_sc_kick(ptr(x), ptr(q), ptr(w), ptr(energy), ptr(length), ptr(grid_extent), mass_eV, B, x.shape[0],
                          q.shape[0], w.shape[0], grid_extent.shape[0], N, b3, dt, ptr(out), ptr(ws), ws_bytes, stream_ptr(),
                          side_stream.cuda_stream if side_stream is not None else None, post_map_ptr, 1), "chx_sc_kick")
    return out


def sc_tile_state(N: int, bins, dtype: torch.dtype, device) -> torch.Tensor | None:
    """State buffer of a chain of tile-ordered kicks (chx_sc_tile_state_bytes), or None when the grid has no tile layout
    (tile edges beyond 16 cells)."""
    nbytes = _lib.lib().chx_sc_tile_state_bytes(N, _bins3(bins), dtype_code(dtype))
    return workspace(nbytes, device) if nbytes else None


def sc_kick_sorted(x, q, w, energy, length, grid_extent, mass_eV, N, bins, state, first: bool, last: bool, side_stream=None,
                   post_map_ptr=None, group=None, index: int = 0) -> torch.Tensor:
    """One kick of a chain on the tile-ordered beam (chx_sc_kick_sorted): x (N,7); q, w (N,) only read when `first`;
    energy, length (1,); grid_extent (1,3). Returns (N,7) in tile order, or in the caller's order when `last`.

    `group` (sharding.particle_sharded): the rows are this rank's slice of the beam. The kick then runs in its two halves
    (chx_sc_kick_sorted_begin / _finish) around the two exchanges of the staged sharded kick — the beam moments (the 29-double
    all-gather + chx_merge_moments: this rank's share comes from chx_moments on the first kick and from the sums the previous
    gather pass left in `state` afterwards) and the all-reduce of the deposited charge grid — and the rows stay in tile order."""
    lib = _lib.lib()
    b3 = _bins3(bins)
    dt = dtype_code(x.dtype)
    ws_bytes = lib.chx_sc_kick_sorted_workspace_bytes(N, b3, dt)
    ws = workspace(ws_bytes, x.device)
    out = torch.empty((N, 7), dtype=x.dtype, device=x.device)
    flags = (1 if first else 0) | (2 if last else 0)
    side = side_stream.cuda_stream if side_stream is not None else None
    if group is None:
        # `index` (the kick's position in its chain; 0 for the first): from the second kick on the grid geometry and the deposit's
        # bookkeeping are formed inside the kernels that need them (csrc/chx_sc_geom_dev.h) — two launches fewer per kick
        flags |= (index & 0x7FFFFF) << 8
        check(lib.chx_sc_kick_sorted(ptr(x), ptr(q), ptr(w), ptr(energy), ptr(length), ptr(grid_extent), mass_eV, N, b3, dt, ptr(out),
                                     ptr(ws), ws_bytes, ptr(state), state.numel(), flags, stream_ptr(), side, post_map_ptr),
              "chx_sc_kick_sorted")
        return out
    from . import sharding

    if first:
        local = _ops._moments_raw(x.reshape(1, N, 7), w.reshape(1, N), 1, N).reshape(1, MOM_NOUT)
    else:
        local = torch.empty((1, MOM_NOUT), dtype=torch.float64, device=x.device)
        check(lib.chx_sc_tile_beam_moments(ptr(state), state.numel(), N, b3, dt, ptr(local), stream_ptr()), "chx_sc_tile_beam_moments")
    mom, rows = sharding.gather_moments_rows(local, group)
    rho_addr = ctypes.c_void_p()
    check(lib.chx_sc_kick_sorted_begin(ptr(x), ptr(q), ptr(w), ptr(energy), ptr(length), ptr(grid_extent), mass_eV, N, b3, dt, ptr(ws),
                                       ws_bytes, ptr(state), state.numel(), flags, ptr(mom), rows, ctypes.byref(rho_addr), stream_ptr(),
                                       side),
          "chx_sc_kick_sorted_begin")
    off = rho_addr.value - state.data_ptr()         # the chain's accumulation grid, inside the state buffer
    rho = state[off:off + int(bins[0]) * int(bins[1]) * int(bins[2]) * x.element_size()].view(x.dtype)
    sharding.allreduce_grid(rho, group)
    check(lib.chx_sc_kick_sorted_finish(ptr(x), ptr(energy), mass_eV, N, b3, dt, ptr(out), ptr(ws), ws_bytes, ptr(state), state.numel(),
                                        flags, stream_ptr(), side, post_map_ptr), "chx_sc_kick_sorted_finish")
    return out


def sc_gather_kick(x, F, half, cell, energy, dt, mass_eV, B, N, bins) -> torch.Tensor:
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_sc_gather_kick(ptr(x), ptr(F), ptr(half), ptr(cell), ptr(energy), ptr(dt), mass_eV, B,
                                        x.shape[0], energy.shape[0], N, _bins3(bins), dtype_code(x.dtype),
                                        ptr(out), stream_ptr()), "chx_sc_gather_kick")
    return out


def sc_gather_kick_phi(x, phi_halo, half, cell, gamma, energy, dt, mass_eV, B, N, bins, post_map=None) -> torch.Tensor:
    """sc_gradient + sc_gather_kick (+ the linear run post_map) in one pass from the potential with a halo
    (chx_sc_gather_kick_phi): no force grid."""
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(_lib.lib().chx_sc_gather_kick_phi(ptr(x), ptr(phi_halo), ptr(half), ptr(cell), ptr(gamma), ptr(energy), ptr(dt), mass_eV,
                                            B, x.shape[0], energy.shape[0], N, _bins3(bins), dtype_code(x.dtype),
                                            ptr(post_map) if post_map is not None else None,
                                            post_map.shape[0] if post_map is not None else 1, ptr(out), stream_ptr()),
          "chx_sc_gather_kick_phi")
    return out


class ScPoisson(torch.autograd.Function):
    """phi = pot_scale * (G(cell, gamma) * rho) on a power-of-two grid (chx_sc_igf_table + chx_sc_green_spectrum +
    chx_sc_convolve). The operator is self-adjoint (G is even), so d rho = the same convolution of d phi; the cell /
    gamma derivatives come from convolving rho with the derivative tables of chx_sc_igf_table_grad."""

    @staticmethod
    def forward(ctx, rho, cell, gamma, pot_scale, bins):
        Ghat = sc_green_spectrum(cell, gamma, bins)
        phi = sc_convolve(rho, Ghat, pot_scale, bins)
        ctx.save_for_backward(rho, cell, gamma, pot_scale, Ghat, phi)
        ctx.bins = tuple(bins)
        return phi

    @staticmethod
    def backward(ctx, dphi):
        rho, cell, gamma, pot_scale, Ghat, phi = ctx.saved_tensors
        bins = ctx.bins
        B = rho.shape[0]
        dphi = dphi.contiguous()
        d_rho = d_cell = d_gamma = d_scale = None
        if ctx.needs_input_grad[0]:
            d_rho = sc_convolve(dphi, Ghat, pot_scale, bins)
        if ctx.needs_input_grad[3]:
            d_scale = (dphi.double() * phi.double()).sum(dim=(1, 2, 3)) / pot_scale
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            lib = _lib.lib()
            b3 = _bins3(bins)
            dt = dtype_code(cell.dtype)
            n1 = (bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1)
            tables = torch.empty((3, B, n1), dtype=torch.float64, device=cell.device)
            check(lib.chx_sc_igf_table_grad(ptr(cell), ptr(gamma), B, b3, dt, ptr(tables), stream_ptr()),
                  "chx_sc_igf_table_grad")
            ws_bytes = lib.chx_sc_green_workspace_bytes(B, b3, dt)
            ws = workspace(ws_bytes, cell.device)
            sens = []
            for d in range(3):
                Gd = torch.empty_like(Ghat)
                check(lib.chx_sc_green_spectrum(ptr(tables[d]), B, b3, dt, ptr(Gd), ptr(ws), ws_bytes, stream_ptr()),
                      "chx_sc_green_spectrum")
                phi_d = sc_convolve(rho, Gd, pot_scale, bins)
                sens.append((dphi.double() * phi_d.double()).sum(dim=(1, 2, 3)))
            # third table is the derivative with respect to cell_z * gamma
            if ctx.needs_input_grad[1]:
                d_cell = torch.stack([sens[0], sens[1], sens[2] * gamma.double()], dim=-1).to(cell.dtype)
            if ctx.needs_input_grad[2]:
                d_gamma = (sens[2] * cell[:, 2].double()).to(gamma.dtype)
        return d_rho, d_cell, d_gamma, d_scale, None


class ScPoissonDense(torch.autograd.Function):
    """The Poisson stage on ANY grid: the same operator as ScPoisson, with hipFFT plans owned by libchx (ScFftPlan) on the
    zero-padded (2g)^3 arrays instead of the pruned power-of-two transforms. Self-adjoint like ScPoisson; the cell / gamma
    derivatives convolve rho with the doubled arrays of the derivative tables (chx_sc_igf_table_grad + chx_sc_igf_from_table)."""

    @staticmethod
    def _convolve(rho, Ghat, pot_scale, bins, plan):
        B, (gx, gy, gz) = rho.shape[0], bins
        work = torch.zeros((B, 2 * gx, 2 * gy, 2 * gz + 2), dtype=rho.dtype, device=rho.device)
        work[:, :gx, :gy, :gz] = rho
        plan.forward(work, which=0)
        sc_spectral_mul(work, Ghat, pot_scale)
        plan.inverse(work)
        return work[:, :gx, :gy, :gz].contiguous()

    @staticmethod
    def forward(ctx, rho, cell, gamma, pot_scale, bins, plan):
        Ghat = sc_igf(cell, gamma, bins, padded=True)
        plan.forward(Ghat, which=1)
        phi = ScPoissonDense._convolve(rho, Ghat, pot_scale, bins, plan)
        ctx.save_for_backward(rho, cell, gamma, pot_scale, Ghat, phi)
        ctx.bins, ctx.plan = tuple(bins), plan
        return phi

    @staticmethod
    def backward(ctx, dphi):
        rho, cell, gamma, pot_scale, Ghat, phi = ctx.saved_tensors
        bins, plan = ctx.bins, ctx.plan
        B = rho.shape[0]
        dphi = dphi.contiguous()
        d_rho = d_cell = d_gamma = d_scale = None
        if ctx.needs_input_grad[0]:
            d_rho = ScPoissonDense._convolve(dphi, Ghat, pot_scale, bins, plan)
        if ctx.needs_input_grad[3]:
            d_scale = (dphi.double() * phi.double()).sum(dim=(1, 2, 3)) / pot_scale
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            lib = _lib.lib()
            b3 = _bins3(bins)
            dt = dtype_code(cell.dtype)
            n1 = (bins[0] + 1) * (bins[1] + 1) * (bins[2] + 1)
            tables = torch.empty((3, B, n1), dtype=torch.float64, device=cell.device)
            check(lib.chx_sc_igf_table_grad(ptr(cell), ptr(gamma), B, b3, dt, ptr(tables), stream_ptr()), "chx_sc_igf_table_grad")
            sens = []
            for d in range(3):
                Gd = torch.empty_like(Ghat)
                check(lib.chx_sc_igf_from_table(ptr(tables[d]), B, b3, dt, ptr(Gd), Ghat.shape[-1], stream_ptr()),
                      "chx_sc_igf_from_table")
                plan.forward(Gd, which=1)
                phi_d = ScPoissonDense._convolve(rho, Gd, pot_scale, bins, plan)
                sens.append((dphi.double() * phi_d.double()).sum(dim=(1, 2, 3)))
            if ctx.needs_input_grad[1]:     # third table: derivative with respect to cell_z * gamma
                d_cell = torch.stack([sens[0], sens[1], sens[2] * gamma.double()], dim=-1).to(cell.dtype)
            if ctx.needs_input_grad[2]:
                d_gamma = (sens[2] * cell[:, 2].double()).to(gamma.dtype)
        return d_rho, d_cell, d_gamma, d_scale, None, None


class ScGradient(torch.autograd.Function):
    """F = -(1/gamma^2) grad phi, packed (B, gx, gy, gz, 4) (chx_sc_gradient / chx_sc_gradient_bwd)."""

    @staticmethod
    def forward(ctx, phi, cell, gamma, bins):
        F = sc_gradient(phi, cell, gamma, bins)
        ctx.save_for_backward(cell, gamma, F)
        ctx.bins = tuple(bins)
        return F

    @staticmethod
    def backward(ctx, dF):
        cell, gamma, F = ctx.saved_tensors
        bins = ctx.bins
        B = F.shape[0]
        dF = dF.contiguous()
        d_phi = d_cell = d_gamma = None
        if ctx.needs_input_grad[0]:
            d_phi = torch.empty((B, *bins), dtype=F.dtype, device=F.device)
            check(_lib.lib().chx_sc_gradient_bwd(ptr(dF), ptr(cell), ptr(gamma), B, _bins3(bins), dtype_code(F.dtype),
                                                 ptr(d_phi), stream_ptr()), "chx_sc_gradient_bwd")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # F_d = -(1/gamma^2) (0.5 / cell_d) (difference): homogeneous of degree -1 in cell_d and -2 in gamma
            dot = (dF.double() * F.double()).sum(dim=(1, 2, 3))[:, :3]      # (B, 3)
            if ctx.needs_input_grad[1]:
                d_cell = (-dot / cell.double()).to(cell.dtype)
            if ctx.needs_input_grad[2]:
                d_gamma = (-2.0 * dot.sum(dim=-1) / gamma.double()).to(gamma.dtype)
        return d_phi, d_cell, d_gamma, None


class ScGatherKick(torch.autograd.Function):
    """Fused SI conversion + trilinear gather + kick (chx_sc_gather_kick); backward = chx_sc_gather_kick_bwd."""

    @staticmethod
    def forward(ctx, x, F, half, cell, energy, dt, mass_eV, B, N, bins):
        ctx.save_for_backward(x, F, half, cell, energy, dt)
        ctx.meta = (mass_eV, B, N, tuple(bins))
        return sc_gather_kick(x, F, half, cell, energy, dt, mass_eV, B, N, bins)

    @staticmethod
    def backward(ctx, dY):
        x, F, half, cell, energy, dt = ctx.saved_tensors
        mass_eV, B, N, bins = ctx.meta
        need = ctx.needs_input_grad
        dY = aligned(dY.contiguous())
        lib = _lib.lib()
        dx = torch.empty((B, N, 7), dtype=x.dtype, device=x.device) if need[0] else None
        dF = torch.zeros_like(F) if need[1] else None
        partials = None
        if need[2] or need[3] or need[4] or need[5]:
            partials = torch.empty((lib.chx_sc_gather_kick_bwd_partials_count(B, N),), dtype=torch.float64, device=x.device)
        check(lib.chx_sc_gather_kick_bwd(ptr(x), ptr(F), ptr(half), ptr(cell), ptr(energy), ptr(dt), ptr(dY), mass_eV, B,
                                         x.shape[0], energy.shape[0], N, _bins3(bins), dtype_code(x.dtype), ptr(dx), ptr(dF),
                                         ptr(partials), stream_ptr()), "chx_sc_gather_kick_bwd")
        d_half = d_cell = d_energy = d_dt = None
        if partials is not None:
            tot = partials.view(B, -1, 8).sum(dim=1)
            d_half = tot[:, 0:3].to(half.dtype) if need[2] else None
            d_cell = tot[:, 3:6].to(cell.dtype) if need[3] else None
            d_dt = tot[:, 6].to(dt.dtype) if need[5] else None
            if need[4]:
                d_energy = tot[:, 7]
                d_energy = (d_energy.sum(dim=0, keepdim=True) if energy.shape[0] == 1 and B > 1 else d_energy).to(energy.dtype)
        if dx is not None and x.shape[0] == 1 and B > 1:
            dx = dx.sum(dim=0, keepdim=True)
        return dx, dF, d_half, d_cell, d_energy, d_dt, None, None, None, None


def _si(fn_name, particles, energy, mass_eV):
    require_device(particles, energy)
    N = particles.shape[-2]
    batch_shape = torch.broadcast_shapes(particles.shape[:-2], energy.shape)
    B = numel(batch_shape)
    x, Bx = flat_bcast(particles, batch_shape, 2)
    e, Be = flat_bcast(energy.to(particles.dtype), batch_shape, 0)
    x, e = aligned(x), e.contiguous()
    out = torch.empty((B, N, 7), dtype=x.dtype, device=x.device)
    check(getattr(_lib.lib(), fn_name)(ptr(x), ptr(e), mass_eV, B, Bx, Be, N, dtype_code(x.dtype), ptr(out),
                                       stream_ptr()), fn_name)
    return out.reshape(*batch_shape, N, 7)


def to_xyz_pxpypz(particles, energy, mass_eV):
    return _si("chx_to_xyz_pxpypz", particles, energy, mass_eV)


def from_xyz_pxpypz(xp, energy, mass_eV):
    return _si("chx_from_xyz_pxpypz", xp, energy, mass_eV)

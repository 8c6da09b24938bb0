"""SpaceChargeKick (mirror of cheetah/accelerator/space_charge_kick.py:54-609).

One kick, all on the caller's stream, no host synchronisation:

* power-of-two grids (16 ... 512 per axis): ONE C call, `chx_sc_kick` — beam sizes and grid geometry
  (`chx_sc_beam_geometry`), sorted LDS-privatised cloud-in-cell deposit, Green spectrum on a side stream
  (`chx_sc_green_spectrum_fast`), libchx's pruned line-FFT convolution into a potential with a halo
  (`chx_sc_convolve_halo`), and one particle pass that takes the field's central differences on the potential around the
  particle's cell, interpolates, kicks and converts to SI and back (`chx_sc_gather_kick_phi`; no force grid);
* other grids: the same stages with dense in-place hipFFT plans owned by libchx (`chx_sc_fft_*`) on the zero-padded
  (2g)^3 Hockney arrays, the Green-function chain on a side stream;
* a beam whose particles are spread over the ranks of a process group (`sharding.particle_sharded`): the staged form
  with the two exchanges in between (global beam moments, summed charge grid);
* anything that requires grad: the differentiable stages of `_track_differentiable`.
"""

from __future__ import annotations

import math

import torch

from .. import _ops
from .._cache import TensorKey
from ..particles.particle_beam import ParticleBeam
from .element import Element

import os as _os

_CHAIN_SIDE_STREAM = _os.environ.get("CHX_SC_CHAIN_SIDE_STREAM", "1") != "0"

epsilon_0 = 8.8541878188e-12  # scipy.constants.epsilon_0 (CODATA 2022)
speed_of_light = 299792458.0


class SpaceChargeKick(Element):
    """Applies the effect of space charge over `effect_length` as an instantaneous momentum kick."""

    _plans: dict = {}
    _side_streams: dict = {}

    @classmethod
    def _fft_plan(cls, B, g, dtype):
        key = (B, tuple(g), dtype)
        plan = cls._plans.get(key)
        if plan is None:
            if len(cls._plans) >= 8:  # bounded: every plan holds hipFFT work areas
                cls._plans.pop(next(iter(cls._plans)))
            plan = cls._plans[key] = _ops.ScFftPlan(B, g, dtype)
        return plan

    @classmethod
    def _chain_side_stream(cls, device):
        """The stream a CHAIN kick puts its Green-function kernels (and the next run's map) on; None = the caller's own
        (CHX_SC_CHAIN_SIDE_STREAM=0: an A/B switch, benchmarks/_c4_side_stream_ab.sh)."""
        return cls._side_stream(device) if _CHAIN_SIDE_STREAM else None

    @classmethod
    def _side_stream(cls, device):
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        s = cls._side_streams.get(key)
        if s is None:
            s = cls._side_streams[key] = torch.cuda.Stream(device=device)
        return s

    def __init__(self, effect_length, grid_shape=(32, 32, 32), grid_extent_x=None, grid_extent_y=None,
                 grid_extent_tau=None, name=None, sanitize_name=None, metadata=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__(name=name, sanitize_name=sanitize_name, metadata=metadata, **fk)
        self.grid_shape = tuple(int(g) for g in grid_shape)
        three = lambda v: v if v is not None else torch.tensor(3.0, **fk)  # noqa: E731
        self.register_buffer_or_parameter("effect_length", effect_length)
        self.register_buffer_or_parameter("grid_extent_x", three(grid_extent_x))
        self.register_buffer_or_parameter("grid_extent_y", three(grid_extent_y))
        self.register_buffer_or_parameter("grid_extent_tau", three(grid_extent_tau))

    @property
    def is_skippable(self) -> bool:
        return False

    def first_order_transfer_map(self, energy, species):
        raise NotImplementedError("SpaceChargeKick has no linear transfer map")

    def _grid_extent(self, dtype) -> torch.Tensor:
        """(Bext, 3) grid extents in sigmas, cached against the three buffers' identity / version."""
        parts = (self.grid_extent_x, self.grid_extent_y, self.grid_extent_tau)
        cached = self.__dict__.get("_ext_cache")
        if cached is None or cached[0] != dtype or not cached[1].matches(parts) or any(p.requires_grad for p in parts):
            ext = torch.stack(torch.broadcast_tensors(*parts), dim=-1).to(dtype).reshape(-1, 3).contiguous()
            cached = (dtype, TensorKey(parts), ext)
            self.__dict__["_ext_cache"] = cached
        return cached[2]

    def track(self, incoming: ParticleBeam) -> ParticleBeam:
        assert isinstance(incoming, ParticleBeam), \
            "SpaceChargeKick tracking is currently only supported for `ParticleBeam`."
        parts = incoming.particles
        if torch.is_grad_enabled() and any(t.requires_grad for t in (
                parts, incoming.particle_charges, incoming.survival_probabilities, incoming.energy, self.effect_length,
                self.grid_extent_x, self.grid_extent_y, self.grid_extent_tau)):
            return self._track_differentiable(incoming)
        dtype, device = parts.dtype, parts.device
        N = parts.shape[-2]
        g = self.grid_shape
        out_shape = _ops.bshapes(parts.shape[:-2], incoming.energy.shape, incoming.particle_charges.shape[:-1],
                                 incoming.survival_probabilities.shape[:-1], self.effect_length.shape)
        B = _ops.numel(out_shape)

        x, _ = _ops.flat_bcast(parts, out_shape, 2)
        x = _ops.aligned(x)
        energy = incoming.energy.to(dtype).expand(out_shape).reshape(B).contiguous()
        q, _ = _ops.flat_bcast(incoming.particle_charges, out_shape, 1)
        w, _ = _ops.flat_bcast(incoming.survival_probabilities, out_shape, 1)
        L = self.effect_length.to(dtype).expand(out_shape).reshape(B)

        from .. import sharding

        group = sharding.active_group()
        if group is not None:
            return self._track_particle_sharded(incoming, group, x, q, w, energy, L, out_shape, B, N)
        if _ops.sc_pruned_supported(g, dtype):
            # the whole kick in one C call (chx_sc_kick): moments, geometry, deposit, libchx's own pruned line FFTs with
            # the Green-function chain on a side stream, field + gather + kick in one particle pass
            out = _ops.sc_kick(x, q.to(dtype).contiguous(), w.to(dtype).contiguous(), energy, L.contiguous(),
                               self._grid_extent(dtype),
                               incoming.species.mass_eV_float, B, N, g, side_stream=self._side_stream(device))
            return ParticleBeam(out.reshape(*out_shape, N, 7), incoming.energy,
                                particle_charges=incoming.particle_charges,
                                survival_probabilities=incoming.survival_probabilities, s=incoming.s,
                                species=incoming.species)

        # beam sizes -> grid geometry (space_charge_kick.py:531-550): one moments call + one geometry kernel
        mom = _ops.moments(x, w.contiguous())                     # (B or 1, 29) float64
        G2 = (2 * g[0], 2 * g[1], 2 * g[2])
        # the inverse FFT below runs unnormalised (norm="forward"): its 1/(8 g^3) goes into the spectral factor
        pot_factor = 1.0 / (4 * math.pi * epsilon_0) / float(G2[0] * G2[1] * G2[2])
        half, cell, gamma, dt, scale, extent, pot_scale = _ops.sc_geometry(
            mom, self._grid_extent(dtype), energy, L.contiguous(), incoming.species.mass_eV_float, pot_factor, B, g)

        # in-place hipFFT on the padded layout [2gx][2gy][2gz+2]; the Green-function chain (fp64-bound table,
        # fill, forward FFT) runs on a side stream while the main stream deposits and transforms the charge
        plan = self._fft_plan(B, g, dtype)
        ldz = G2[2] + 2
        main = torch.cuda.current_stream(device)
        side = self._side_stream(device)
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            green = _ops.sc_igf(cell, gamma, g, padded=True)
            plan.forward(green, which=1)
            join = torch.cuda.Event()
            join.record(side)
        rho = torch.zeros((B, G2[0], G2[1], ldz), dtype=dtype, device=device)
        _ops.cic_deposit_into(rho, (G2[1] * ldz, ldz, 1), G2[0] * G2[1] * ldz, x, (0, 2, 4), g, extent,
                              charge=q, survival=w, scale=scale)
        plan.forward(rho, which=0)
        main.wait_event(join)
        green.record_stream(main)
        _ops.sc_spectral_mul(rho, green, pot_scale)
        plan.inverse(rho)
        phi = rho
        force = _ops.sc_gradient(phi, cell, gamma, g)
        out = _ops.sc_gather_kick(x, force, half, cell, energy, dt, incoming.species.mass_eV_float, B, N, g)
        return ParticleBeam(out.reshape(*out_shape, N, 7), incoming.energy,
                            particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=incoming.s,
                            species=incoming.species)

    def _track_then_map(self, incoming: ParticleBeam, post_map_ptr: int):
        """The kick followed by a linear map (device address of a (7,7) array of the beam dtype) in the SAME particle pass
        (`chx_sc_kick` with `post_map`): what `Segment.track` uses when a run of linear elements follows the kick. Returns the
        tracked particle tensor, or None when the one-call kick does not apply (vectorised beam, gradients, a grid that is not
        a power of two, a particle-sharded beam) and the caller tracks kick and run separately."""
        from .. import sharding

        parts = incoming.particles
        dtype, device = parts.dtype, parts.device
        g = self.grid_shape
        if parts.dim() != 2 or incoming.energy.dim() != 0 or incoming.particle_charges.dim() != 1 \
                or incoming.survival_probabilities.dim() != 1 or self.effect_length.dim() != 0:
            return None
        if sharding.active_group() is not None or not _ops.sc_pruned_supported(g, dtype):
            return None
        if torch.is_grad_enabled() and any(t.requires_grad for t in (
                parts, incoming.particle_charges, incoming.survival_probabilities, incoming.energy, self.effect_length,
                self.grid_extent_x, self.grid_extent_y, self.grid_extent_tau)):
            return None
        N = parts.shape[0]
        x = _ops.aligned(parts).reshape(1, N, 7)
        out = _ops.sc_kick(x, incoming.particle_charges.to(dtype).reshape(1, N).contiguous(),
                           incoming.survival_probabilities.to(dtype).reshape(1, N).contiguous(),
                           incoming.energy.to(dtype).reshape(1), self.effect_length.to(dtype).reshape(1), self._grid_extent(dtype),
                           incoming.species.mass_eV_float, 1, N, g, side_stream=self._side_stream(device),
                           post_map_ptr=post_map_ptr)
        return out.reshape(N, 7)

    def _chain_settings_ok(self, dtype) -> bool:
        """Can this element be a link of a chain of tile-ordered kicks (`Segment.track`)? Scalar settings without gradients
        on a grid libchx's pruned solver takes."""
        if self.effect_length.dim() != 0 or not _ops.sc_pruned_supported(self.grid_shape, dtype):
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in (
            self.effect_length, self.grid_extent_x, self.grid_extent_y, self.grid_extent_tau)))

    def _chain_beam_ok(self, incoming: ParticleBeam) -> bool:
        """One plain beam (no vector dims, no gradients) that is large enough for the tile sort. A particle-sharded beam
        qualifies with its local rows (`_track_in_chain` puts the two exchanges between the halves of the kick)."""
        parts = incoming.particles
        if parts.dim() != 2 or incoming.energy.dim() != 0 or incoming.particle_charges.dim() != 1 \
                or incoming.survival_probabilities.dim() != 1 or parts.shape[0] < _ops.SORTED_CIC_MIN_PARTICLES:
            return False
        return not (torch.is_grad_enabled() and any(t.requires_grad for t in (
            parts, incoming.particle_charges, incoming.survival_probabilities, incoming.energy)))

    def _track_in_chain(self, incoming: ParticleBeam, state: torch.Tensor, first: bool, last: bool, post_map_ptr=None, index: int = 0):
        """This kick as a link of a chain (`chx_sc_kick_sorted`): the first link sorts the particle rows by deposit tile into
        `state`, later links work on the ordered rows, the last one returns the rows in the caller's order. Returns the
        particle tensor (in tile order unless `last`)."""
        from .. import sharding

        parts = incoming.particles
        dtype, device = parts.dtype, parts.device
        N = parts.shape[0]
        x = _ops.aligned(parts)
        q = w = None
        if first:
            q = incoming.particle_charges.to(dtype).contiguous()
            w = incoming.survival_probabilities.to(dtype).contiguous()
        # inside sharding.particle_sharded: the same two exchanges per kick as `_track_particle_sharded` (29 doubles all-gather,
        # grid all-reduce), with the same contents — so a rank may be on the chain while another (a slice below the sort's
        # minimum, a guard that switched its plan back) tracks kick by kick, and both see the same grid
        return _ops.sc_kick_sorted(x, q, w, incoming.energy.to(dtype).reshape(1), self.effect_length.to(dtype).reshape(1),
                                   self._grid_extent(dtype), incoming.species.mass_eV_float, N, self.grid_shape, state, first, last,
                                   side_stream=self._chain_side_stream(device), post_map_ptr=post_map_ptr,
                                   group=sharding.active_group(), index=index)

    def _track_particle_sharded(self, incoming, group, x, q, w, energy, L, out_shape, B, N) -> ParticleBeam:
        """The kick for a beam whose particles are spread over the ranks of `group` (sharding.particle_sharded): the same
        stages as `chx_sc_kick`, issued one by one so that the two exchanges fit in between — the beam moments (grid
        geometry from the GLOBAL sigmas) and the charge grid (sum over the shards). Every rank then solves the same
        Poisson problem and kicks its own particles. Any grid shape: power-of-two grids take libchx's pruned line FFTs,
        the others the dense hipFFT plans of the single-GPU path (space_charge_kick.py:57,125-161)."""
        from .. import sharding

        dtype, device = x.dtype, x.device
        g = self.grid_shape
        pruned = _ops.sc_pruned_supported(g, dtype)     # else: dense hipFFT plans on the zero-padded (2g)^3 arrays, any grid
        mass = incoming.species.mass_eV_float
        # exchange 1: beam moments (29 doubles per rank and batch row, one all-gather + chx_merge_moments)
        mom = _ops.moments(x, w.to(dtype).contiguous()).reshape(-1, _ops.MOM_NOUT)
        mom = sharding.gather_merge_moments(mom, group).contiguous()
        pot_factor = 1.0 / (4 * math.pi * epsilon_0) / float(8 * g[0] * g[1] * g[2])
        half, cell, gamma, dt, scale, extent, pot_scale = _ops.sc_geometry(
            mom, self._grid_extent(dtype), energy, L.contiguous(), mass, pot_factor, B, g)
        # the Green-function chain depends on the geometry only: it runs on the side stream UNDER the deposit and the grid
        # all-reduce (the main stream joins it in front of the convolution)
        main = torch.cuda.current_stream(device)
        side = self._side_stream(device)
        fork = torch.cuda.Event()
        fork.record(main)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            if pruned:
                green = _ops.sc_green_spectrum(cell, gamma, g)
            else:
                plan = self._fft_plan(B, g, dtype)
                green = _ops.sc_igf(cell, gamma, g, padded=True)
                plan.forward(green, which=1)
            join = torch.cuda.Event()
            join.record(side)
        # local charge: the tile-sorted deposit stores every cell of the compact grid itself (no zero-fill pass)
        rho = _ops.sc_deposit_overwrite(x, q, w, extent, scale, B, N, g)
        # exchange 2: the charge grid summed over the shards (g^3 values; 8.4 MB at 128^3 fp32)
        sharding.allreduce_grid(rho, group)
        main.wait_event(join)
        green.record_stream(main)
        if pruned:
            phi = _ops.sc_convolve_halo(rho, green, pot_scale, g)
            out = _ops.sc_gather_kick_phi(x, phi, half, cell, gamma, energy, dt, mass, B, N, g)
        else:
            work = torch.zeros((B, 2 * g[0], 2 * g[1], 2 * g[2] + 2), dtype=dtype, device=device)
            work[:, :g[0], :g[1], :g[2]] = rho
            plan.forward(work, which=0)
            _ops.sc_spectral_mul(work, green, pot_scale)
            plan.inverse(work)
            force = _ops.sc_gradient(work, cell, gamma, g)
            out = _ops.sc_gather_kick(x, force, half, cell, energy, dt, mass, B, N, g)
        return ParticleBeam(out.reshape(*out_shape, N, 7), incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=incoming.s,
                            species=incoming.species)

    def _track_differentiable(self, incoming: ParticleBeam) -> ParticleBeam:
        """The same pipeline as `track`, assembled from differentiable stages (Moments, CicDeposit, ScPoisson,
        ScGradient, ScGatherKick); the (B,)-sized grid geometry is written as tensor expressions so that autograd links
        the stages exactly like the reference's tensor code does (space_charge_kick.py:531-575)."""
        parts = incoming.particles
        dtype, device = parts.dtype, parts.device
        g = self.grid_shape
        pruned = _ops.sc_pruned_supported(g, dtype)     # else: the dense hipFFT plans of the forward-only path (ScPoissonDense)
        N = parts.shape[-2]
        out_shape = _ops.bshapes(parts.shape[:-2], incoming.energy.shape, incoming.particle_charges.shape[:-1],
                                 incoming.survival_probabilities.shape[:-1], self.effect_length.shape)
        B = _ops.numel(out_shape)
        x = parts.expand(*out_shape, N, 7).reshape(B, N, 7).contiguous()
        energy = incoming.energy.to(dtype).expand(out_shape).reshape(B).contiguous()
        q = incoming.particle_charges.expand(*out_shape, N).reshape(B, N)
        w = incoming.survival_probabilities.expand(*out_shape, N).reshape(B, N)
        L = self.effect_length.to(dtype).expand(out_shape).reshape(B)

        mom = _ops.moments(x, w.contiguous())                                   # (B, 29) float64, differentiable
        sigma = mom[:, [8, 19, 26]].sqrt().to(dtype)                            # sigma_x, sigma_y, sigma_tau
        half = self._grid_extent(dtype) * sigma                                 # (B, 3)
        gt = self.__dict__.get("_grid_tensor")       # the three grid sizes on the device, made once (a host-to-device copy)
        if gt is None or gt[0] != (tuple(g), dtype, device):
            gt = self.__dict__["_grid_tensor"] = ((tuple(g), dtype, device),
                                                  torch.tensor([float(v) for v in g], dtype=dtype, device=device))
        gt = gt[1]
        cell = 2 * half / gt
        gamma = energy / incoming.species.mass_eV_float
        ig2 = 1 / (gamma * gamma)
        beta = torch.where(gamma.abs() > 0, (1 - ig2).clamp_min(0).sqrt(), torch.ones_like(gamma))
        dt = L / (speed_of_light * beta)
        scale = torch.stack([torch.ones_like(beta), torch.ones_like(beta), -beta], dim=-1)
        extent = torch.stack([-half, half], dim=-1)
        G2 = (2 * g[0], 2 * g[1], 2 * g[2])
        pot_factor = 1.0 / (4 * math.pi * epsilon_0) / float(G2[0] * G2[1] * G2[2])
        pot_scale = pot_factor / cell.double().prod(dim=-1)

        rho = _ops.cic_deposit(x, (0, 2, 4), g, extent, charge=q, survival=w, scale=scale)
        if pruned:
            phi = _ops.ScPoisson.apply(rho.reshape(B, *g).contiguous(), cell.contiguous(), gamma.contiguous(), pot_scale, g)
        else:
            phi = _ops.ScPoissonDense.apply(rho.reshape(B, *g).contiguous(), cell.contiguous(), gamma.contiguous(), pot_scale, g,
                                            self._fft_plan(B, g, dtype))
        force = _ops.ScGradient.apply(phi, cell.contiguous(), gamma.contiguous(), g)
        out = _ops.ScGatherKick.apply(_ops.aligned(x), force, half.contiguous(), cell.contiguous(), energy, dt.contiguous(),
                                      incoming.species.mass_eV_float, B, N, g)
        return ParticleBeam(out.reshape(*out_shape, N, 7), incoming.energy, particle_charges=incoming.particle_charges,
                            survival_probabilities=incoming.survival_probabilities, s=incoming.s,
                            species=incoming.species)

    @property
    def defining_features(self) -> list[str]:
        return super().defining_features + ["effect_length", "grid_shape", "grid_extent_x", "grid_extent_y",
                                            "grid_extent_tau"]

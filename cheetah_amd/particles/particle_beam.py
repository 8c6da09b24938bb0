"""ParticleBeam (mirror of cheetah/particles/particle_beam.py:60-106, :1262-1346, :1672-2001 and the
derived optics of cheetah/particles/beam.py:323-556).

State layout is the reference's: ``particles (…, N, 7)``, ``energy (…)``, ``particle_charges (…, N)``,
``survival_probabilities (…, N)``, ``s``, ``species``. All beam moments come from ONE fused HIP
reduction (`chx_moments`: 6 means + 21 covariances in two passes over the particle array) and are
cached per (tensor identity, version); the reference re-reads a strided column 3-4 times for every
single property.
"""

from __future__ import annotations

import math

import torch

from .. import _ops
from .._cache import TensorKey
from ..sharding import _ACTIVE_GROUP as _SHARDING_STACK       # non-empty inside `sharding.particle_sharded(...)`
from .beam import Beam
from .species import Species

_COORDS = ["x", "px", "y", "py", "tau", "p"]
speed_of_light = 299792458.0


def _tri(i: int, j: int) -> int:
    if i > j:
        i, j = j, i
    return 8 + i * 6 - (i * (i - 1)) // 2 + (j - i)


class _MomentEntry(torch.autograd.Function):
    """mom[..., index] (or its square root) cast to `dtype`; backward scatters the incoming gradient into that entry."""

    @staticmethod
    def forward(ctx, mom, index, take_sqrt, dtype):
        v = mom[..., index]
        if take_sqrt:
            v = v.sqrt()
            ctx.save_for_backward(v)
        ctx.index, ctx.take_sqrt, ctx.mom_shape = index, take_sqrt, mom.shape
        return v.to(dtype)

    @staticmethod
    def backward(ctx, grad):
        g = grad.to(torch.float64)
        if ctx.take_sqrt:
            (v,) = ctx.saved_tensors
            g = g * 0.5 / v                      # d sqrt(x) / dx, infinite at x = 0 like torch.sqrt's own backward
        d = g.new_zeros(ctx.mom_shape)
        d[..., ctx.index] = g
        return d, None, None, None


class ParticleBeam(Beam):
    """Beam of macro-particles, each a 7-vector (x, px, y, py, tau, p, 1)."""

    PRETTY_DIMENSION_LABELS = {"x": r"$x$", "px": r"$p_x$", "y": r"$y$", "py": r"$p_y$", "tau": r"$\tau$", "p": r"$\delta$"}
    UNVECTORIZED_NUM_ATTR_DIMS = Beam.UNVECTORIZED_NUM_ATTR_DIMS | {
        "particles": 2, "particle_charges": 1, "survival_probabilities": 1, "x": 1, "px": 1, "y": 1, "py": 1, "tau": 1, "p": 1}

    def _no_plot(self, *args, **kwargs):
        from .beam import _OUT_OF_SCOPE

        raise NotImplementedError(_OUT_OF_SCOPE.format("ParticleBeam.plot_*"))

    plot_1d_distribution = plot_2d_distribution = plot_distribution = plot_point_cloud = _no_plot

    def __init__(self, particles, energy, particle_charges=None, survival_probabilities=None, s=None,
                 species=None, device=None, dtype=None) -> None:
        super().__init__()
        assert particles.shape[-2] > 0 and particles.shape[-1] == 7, "Particle vectors must be 7-dimensional."
        # NOTE: like the reference, device/dtype only apply to tensors created here
        device = device if device is not None else particles.device
        dtype = dtype if dtype is not None else particles.dtype
        factory_kwargs = {"device": device, "dtype": dtype}
        species = species if species is not None else Species("electron", **factory_kwargs)
        self._modules["species"] = species
        # A new beam object is created for every tracked element: the buffers are entered directly
        # (same result as five `register_buffer` calls, ~10 us cheaper) and read back through class-level
        # properties below instead of nn.Module.__getattr__.
        buffers = self._buffers
        buffers["particles"] = particles
        buffers["energy"] = energy
        buffers["particle_charges"] = (
            particle_charges if particle_charges is not None
            else torch.full((particles.shape[-2],), species.num_elementary_charges_float * 1.602176634e-19,
                            **factory_kwargs))
        buffers["survival_probabilities"] = (
            survival_probabilities if survival_probabilities is not None
            else torch.ones(particles.shape[-2], **factory_kwargs))
        buffers["s"] = s if s is not None else torch.tensor(0.0, **factory_kwargs)

    # ------------------------------------------------------------------ factories (input generation)
    @classmethod
    def from_distribution(cls, mu, cov, num_particles=100_000, energy=None, total_charge=None, s=None,
                          species=None, device=None, dtype=None) -> "ParticleBeam":
        """Random particles matched to (mu, cov) (particle_beam.py:355-432). Input factory: generated
        with torch's RNG on `device`; not part of the tracking hot path."""
        factory_kwargs = {"device": device, "dtype": dtype}
        species = species.to(**factory_kwargs) if species is not None else Species("electron", **factory_kwargs)
        energy = energy if energy is not None else torch.tensor(1e8, **factory_kwargs)
        total_charge = total_charge if total_charge is not None else species.charge_coulomb * num_particles
        particle_charges = (torch.ones((*total_charge.shape, num_particles), **factory_kwargs)
                            * total_charge.unsqueeze(-1) / num_particles)
        z = torch.randn(num_particles, 6, **factory_kwargs)
        # whiten the sample, then colour it with chol(cov) and shift to mu (utils/statistics.py:91-143)
        z = z - z.mean(dim=0, keepdim=True)
        c = (z.mT @ z) / (num_particles - 1)
        z = torch.linalg.solve_triangular(torch.linalg.cholesky(c), z.mT, upper=False).mT
        chol = torch.linalg.cholesky(cov + torch.eye(6, **factory_kwargs) * torch.finfo(cov.dtype).tiny)
        p6 = z @ chol.mT + mu.unsqueeze(-2)
        p7 = torch.cat([p6, torch.ones_like(p6[..., :1])], dim=-1)
        return cls(p7, energy, particle_charges=particle_charges, s=s, species=species, device=device, dtype=dtype)

    @classmethod
    def from_parameters(cls, num_particles=100_000, mu_x=None, mu_px=None, mu_y=None, mu_py=None, mu_tau=None,
                        mu_p=None, sigma_x=None, sigma_px=None, sigma_y=None, sigma_py=None, sigma_tau=None,
                        sigma_p=None, cov_xpx=None, cov_ypy=None, cov_taup=None, cov_xp=None, cov_pxp=None, cov_yp=None,
                        cov_pyp=None, cov_xy=None, cov_xpy=None, cov_xtau=None, cov_pxy=None, cov_pxpy=None,
                        cov_pxtau=None, cov_ytau=None, cov_pytau=None, energy=None, total_charge=None, s=None,
                        species=None, device=None, dtype=None) -> "ParticleBeam":
        """Gaussian beam from its means, sigmas and any of the 15 covariances; defaults as particle_beam.py:108-353."""
        fk = {"device": device, "dtype": dtype}
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        mus = torch.broadcast_tensors(d(mu_x, 0.0), d(mu_px, 0.0), d(mu_y, 0.0), d(mu_py, 0.0), d(mu_tau, 0.0),
                                      d(mu_p, 0.0))
        mean = torch.stack(mus, dim=-1)
        offdiag = {(0, 1): cov_xpx, (0, 2): cov_xy, (0, 3): cov_xpy, (0, 4): cov_xtau, (0, 5): cov_xp, (1, 2): cov_pxy,
                   (1, 3): cov_pxpy, (1, 4): cov_pxtau, (1, 5): cov_pxp, (2, 3): cov_ypy, (2, 4): cov_ytau, (2, 5): cov_yp,
                   (3, 4): cov_pytau, (3, 5): cov_pyp, (4, 5): cov_taup}
        sigmas = [d(sigma_x, 175e-6), d(sigma_px, 4e-6), d(sigma_y, 175e-6), d(sigma_py, 4e-6), d(sigma_tau, 8e-6),
                  d(sigma_p, 2e-3)]
        given = {k: v for k, v in offdiag.items() if v is not None}
        shape = torch.broadcast_shapes(*[t.shape for t in sigmas], *[v.shape for v in given.values()])
        cov = torch.zeros(*shape, 6, 6, **fk)
        for i, sg in enumerate(sigmas):
            cov[..., i, i] = sg.square()
        for (i, j), c in given.items():
            cov[..., i, j] = c
            cov[..., j, i] = c
        return cls.from_distribution(mean, cov, num_particles=num_particles, energy=energy,
                                     total_charge=total_charge, s=s, species=species, device=device, dtype=dtype)

    @classmethod
    def from_twiss(cls, num_particles=100_000, beta_x=None, alpha_x=None, emittance_x=None, beta_y=None,
                   alpha_y=None, emittance_y=None, dispersion_x=None, dispersion_px=None, dispersion_y=None,
                   dispersion_py=None, energy=None, sigma_tau=None, sigma_p=None, cov_taup=None, total_charge=None,
                   s=None, species=None, device=None, dtype=None) -> "ParticleBeam":
        """Gaussian beam from Twiss parameters and dispersion (particle_beam.py:434-560)."""
        fk = {"device": device, "dtype": dtype}
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        beta_x, alpha_x, emittance_x = d(beta_x, 0.0), d(alpha_x, 0.0), d(emittance_x, 7.1971891e-13)
        beta_y, alpha_y, emittance_y = d(beta_y, 0.0), d(alpha_y, 0.0), d(emittance_y, 7.1971891e-13)
        dx, dpx, dy, dpy = d(dispersion_x, 0.0), d(dispersion_px, 0.0), d(dispersion_y, 0.0), d(dispersion_py, 0.0)
        sigma_tau, sigma_p, cov_taup = d(sigma_tau, 1e-6), d(sigma_p, 1e-6), d(cov_taup, 0.0)
        sp2 = sigma_p.square()
        return cls.from_parameters(
            num_particles=num_particles,
            sigma_x=(beta_x * emittance_x + dx.square() * sp2).sqrt(),
            sigma_px=(emittance_x * (1 + alpha_x.square()) / beta_x + dpx.square() * sp2).sqrt(),
            sigma_y=(beta_y * emittance_y + dy.square() * sp2).sqrt(),
            sigma_py=(emittance_y * (1 + alpha_y.square()) / beta_y + dpy.square() * sp2).sqrt(),
            sigma_tau=sigma_tau, sigma_p=sigma_p, cov_xpx=-emittance_x * alpha_x + dx * dpx * sp2,
            cov_ypy=-emittance_y * alpha_y + dy * dpy * sp2, cov_taup=cov_taup, cov_xp=dx * sp2, cov_pxp=dpx * sp2,
            cov_yp=dy * sp2, cov_pyp=dpy * sp2, energy=d(energy, 1e8), total_charge=total_charge, s=s, species=species,
            device=device, dtype=dtype)

    @classmethod
    def uniform_3d_ellipsoid(cls, num_particles=100_000, radius_x=None, radius_y=None, radius_tau=None,
                             sigma_px=None, sigma_py=None, sigma_p=None, energy=None, total_charge=None, s=None,
                             species=None, device=None, dtype=None) -> "ParticleBeam":
        """Water-bag ellipsoid (particle_beam.py:562-662)."""
        fk = {"device": device, "dtype": dtype}
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        radius_x, radius_y, radius_tau = d(radius_x, 1e-3), d(radius_y, 1e-3), d(radius_tau, 1e-3)
        beam = cls.from_parameters(num_particles=num_particles, sigma_x=radius_x, sigma_px=sigma_px,
                                   sigma_y=radius_y, sigma_py=sigma_py, sigma_tau=radius_tau, sigma_p=sigma_p,
                                   energy=energy, total_charge=total_charge, s=s, species=species, device=device,
                                   dtype=dtype)
        vs = beam.particles.shape[:-2]
        r = torch.rand(*vs, num_particles, **fk).pow(1 / 3)
        theta = (2 * torch.rand(*vs, num_particles, **fk) - 1).arccos()
        phi = torch.rand(*vs, num_particles, **fk) * 2 * math.pi
        parts = beam.particles.clone()
        parts[..., 0] = r * theta.sin() * phi.cos() * radius_x.unsqueeze(-1)
        parts[..., 2] = r * theta.sin() * phi.sin() * radius_y.unsqueeze(-1)
        parts[..., 4] = r * theta.cos() * radius_tau.unsqueeze(-1)
        beam.particles = parts
        return beam

    # ------------------------------------------------------------------ SI conversion (HIP)
    # ---- import / export of other codes' particle files (particle_beam.py:834-1033) --------------------------
    @classmethod
    def from_astra(cls, path: str, device=None, dtype=None) -> "ParticleBeam":
        """Load an Astra particle distribution (converters/astra.py)."""
        from ..converters.astra import from_astrabeam

        kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
        coordinates, energy, charges = from_astrabeam(path)
        particles = torch.ones((coordinates.shape[0], 7), **kw)
        particles[:, :6] = torch.as_tensor(coordinates, **kw)
        return cls(particles=particles, energy=torch.as_tensor(energy, **kw), particle_charges=torch.as_tensor(charges, **kw),
                   species=Species("electron", **kw), **kw)

    @classmethod
    def from_ocelot(cls, parray, device=None, dtype=None) -> "ParticleBeam":
        """From an Ocelot `ParticleArray` — any object with `rparticles` (6, N), `E` [GeV] and `q_array` [C]
        (particle_beam.py:804-832)."""
        kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
        particles = torch.ones((parray.rparticles.shape[1], 7), **kw)
        particles[:, :6] = torch.as_tensor(parray.rparticles.transpose(), **kw)
        return cls(particles=particles, energy=1e9 * torch.as_tensor(parray.E, **kw),
                   particle_charges=torch.as_tensor(parray.q_array, **kw), species=Species("electron", **kw), **kw)

    @classmethod
    def from_elegant(cls, file_path, device=None, dtype=None) -> "ParticleBeam":
        """Load an Elegant SDDS particle file (needs the `sdds` package, like the reference)."""
        from pathlib import Path

        from ..converters import elegant

        kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
        particles, energy, charges = elegant.convert_beam(Path(file_path), **kw)
        return cls(particles=particles, energy=energy, particle_charges=charges, species=Species("electron", **kw), **kw)

    @classmethod
    def from_openpmd_file(cls, path: str, energy: torch.Tensor, device=None, dtype=None) -> "ParticleBeam":
        """Load an openPMD particle-group HDF5 file (needs openPMD-beamphysics, like the reference)."""
        try:
            import pmd_beamphysics as openpmd
        except ImportError:
            raise ImportError("To use the openPMD beam import, openPMD-beamphysics must be installed.")
        return cls.from_openpmd_particlegroup(openpmd.ParticleGroup(path), energy, device=device, dtype=dtype)

    @classmethod
    def from_openpmd_particlegroup(cls, particle_group, energy: torch.Tensor, device=None, dtype=None) -> "ParticleBeam":
        """From an openPMD `ParticleGroup` — or any object with its attributes x, y, px, py [eV/c], t [s], energy [eV],
        weight [C], status, species (particle_beam.py:918-962)."""
        kw = {"device": device, "dtype": dtype}
        species = Species(particle_group.species, **kw)
        energy = torch.as_tensor(energy, **kw)
        p0c = (energy.square() - species.mass_eV.square()).sqrt()
        col = lambda v: torch.as_tensor(v, **kw)  # noqa: E731
        x, y = col(particle_group.x), col(particle_group.y)
        particles = torch.stack([x, col(particle_group.px) / p0c, y, col(particle_group.py) / p0c,
                                 col(particle_group.t) * 299792458.0, (col(particle_group.energy) - energy) / p0c,
                                 torch.ones_like(x)], dim=-1)
        return cls(particles=particles, energy=energy, particle_charges=col(particle_group.weight),
                   survival_probabilities=col(particle_group.status), species=species, **kw)

    def openpmd_data(self) -> dict:
        """The dictionary the reference hands to `pmd_beamphysics.ParticleGroup(data=...)` (particle_beam.py:1009-1031):
        momenta in eV/c, t = tau / c, boolean status from survival_probabilities > 0.5."""
        if self.particles.dim() != 2:
            raise ValueError("Only non-vectorised particle distributions are supported.")
        px, py = self.px * self.p0c, self.py * self.p0c
        p_total = (self.energies.square() - self.species.mass_eV.square()).sqrt()
        pz = (p_total.square() - px.square() - py.square()).sqrt()
        host = lambda v: v.detach().cpu().numpy()  # noqa: E731
        return {"x": host(self.x), "y": host(self.y), "z": host(self.tau), "px": host(px), "py": host(py), "pz": host(pz),
                "t": host(self.tau / 299792458.0), "weight": host(self.particle_charges),
                "status": host((self.survival_probabilities > 0.5).int()), "species": self.species.name}

    def to_openpmd_particlegroup(self):
        try:
            import pmd_beamphysics as openpmd
        except ImportError:
            raise ImportError("To use the openPMD beam export, openPMD-beamphysics must be installed.")
        return openpmd.ParticleGroup(data=self.openpmd_data())

    def save_as_openpmd_h5(self, path: str) -> None:
        self.to_openpmd_particlegroup().write(path)

    def to_xyz_pxpypz(self) -> torch.Tensor:
        """(x, Px, y, Py, z, Pz, 1) in SI units (particle_beam.py:1316-1346) via `chx_to_xyz_pxpypz`."""
        return _ops.to_xyz_pxpypz(self.particles, self.energy, self.species.mass_eV_float)

    @classmethod
    def from_xyz_pxpypz(cls, xp_coordinates, energy, particle_charges=None, survival_probabilities=None, s=None,
                        species=None, device=None, dtype=None) -> "ParticleBeam":
        """particle_beam.py:1262-1314 via `chx_from_xyz_pxpypz`."""
        sp = species if species is not None else Species("electron", device=xp_coordinates.device,
                                                         dtype=xp_coordinates.dtype)
        parts = _ops.from_xyz_pxpypz(xp_coordinates, energy, sp.mass_eV_float)
        return cls(parts, energy, particle_charges=particle_charges, survival_probabilities=survival_probabilities,
                   s=s, species=sp, device=device, dtype=dtype)

    # ------------------------------------------------------------------ moments (HIP, cached)
    def _moments(self) -> torch.Tensor:
        p, w = self.particles, self.survival_probabilities
        if _SHARDING_STACK:
            from .. import sharding

            group = sharding.active_group()
            if group is not None:
                return self._global_moments(group)
        cached = self.__dict__.get("_moment_cache")
        if cached is not None and cached[0].matches((p, w)) and not p.requires_grad and not w.requires_grad:
            return cached[1]
        out = _ops.moments(p, w)
        self.__dict__["_moment_cache"] = (TensorKey((p, w)), out)
        return out

    def _global_moments(self, group) -> torch.Tensor:
        """The moments of ALL shards of a particle-sharded beam (inside `sharding.particle_sharded`): this rank's one-pass reduction,
        ONE all-gather of 29 doubles per rank and batch row, the exact merge (`chx_merge_moments`; utils/statistics.py:4-62 over the
        union of the shards) — once per version of the beam's tensors, whatever number of properties is read
        (particle_beam.py:1699-1943: `sigma_x`, `mu_*`, `cov_*`, emittances, Twiss). A COLLECTIVE: every rank of the group must
        read a property of its shard at the same point of the program. With gradients enabled and particles / weights that carry
        a graph the result is differentiable (`_ops.Moments` with the group): every rank's backward pass works on its own rows."""
        from .. import sharding

        p, w = self.particles, self.survival_probabilities
        if torch.is_grad_enabled() and (p.requires_grad or w.requires_grad):
            # differentiable: one node whose backward is chx_moments_bwd_w on this rank's rows with the GLOBAL moments (no
            # collective); a rank's backward pass yields its shard's share of the gradient of a replicated setting
            # (sharding.all_reduce_gradients sums the shares). Not cached: the result carries this call's graph.
            return _ops.moments(p, w, group=group)
        cached = self.__dict__.get("_global_moment_cache")
        if cached is not None and cached[0].matches((p, w)) and cached[1] is group:
            return cached[2]
        with torch.no_grad():
            local = _ops.moments(p, w)
            out = sharding.gather_merge_moments(local.reshape(-1, _ops.MOM_NOUT), group).reshape(local.shape)
        self.__dict__["_global_moment_cache"] = (TensorKey((p, w)), group, out)
        return out

    def as_parameter_beam(self):
        """ParameterBeam with this beam's means and covariances (particle_beam.py `as_parameter_beam`): one fused
        chx_moments call instead of 27 separate reductions."""
        from .parameter_beam import ParameterBeam

        # like the reference (particle_beam.py:1168-1178) neither `s` nor the species is passed on: `s` restarts at 0 and the
        # result is an ELECTRON beam whatever this beam's species is (a quirk that is kept: same results on the same calls)
        return ParameterBeam._from_moment_vector(self._moments(), self.particles.dtype, self.energy,
                                                 total_charge=self.total_charge)

    def _as_parameter_beam_same_species(self):
        """`as_parameter_beam` with this beam's species and path length (for the engine's own fused observables)."""
        from .parameter_beam import ParameterBeam

        return ParameterBeam._from_moment_vector(self._moments(), self.particles.dtype, self.energy,
                                                 total_charge=self.total_charge, s=self.s, species=self.species)

    def _entry(self, index: int, take_sqrt: bool = False) -> torch.Tensor:
        """One entry of the moment vector (optionally its square root) in the beam's dtype. Under autograd this is ONE node
        (`_MomentEntry`) instead of select -> sqrt -> to, whose three backward nodes cost more host time than the moment
        kernels themselves in an optimisation loop."""
        p = self.particles
        sharded = False
        if _SHARDING_STACK:
            from .. import sharding

            sharded = sharding.active_group() is not None
        if p.requires_grad and not sharded and getattr(p, "_chx_lin", None) is not None:
            # a linearly tracked beam whose only differentiable input is the map: one node, algebraic backward
            v = _ops.moment_entry(p, self.survival_probabilities, index, take_sqrt)
            if v is not None:
                return v
        mom = self._moments()
        if mom.requires_grad and torch.is_grad_enabled():
            return _MomentEntry.apply(mom, index, take_sqrt, self.particles.dtype)
        v = mom[..., index]
        return (v.sqrt() if take_sqrt else v).to(self.particles.dtype)

    def _mu(self, i: int) -> torch.Tensor:
        return self._entry(2 + i)

    def _cov(self, i: int, j: int) -> torch.Tensor:
        return self._entry(_tri(i, j))

    def _sigma(self, i: int) -> torch.Tensor:
        return self._entry(_tri(i, i), take_sqrt=True)

    @property
    def total_charge(self) -> torch.Tensor:
        """Sum of charge x survival probability. Inside `sharding.particle_sharded`: of ALL shards — a COLLECTIVE like the moment
        properties (every rank reads it at the same point; `local_total_charge` is this rank's own); a graph on this rank's
        charges / survival probabilities is kept (`sharding.sum_over_ranks`)."""
        q = self.local_total_charge
        if _SHARDING_STACK:
            from .. import sharding

            group = sharding.active_group()
            if group is not None:
                q = sharding.sum_over_ranks(q.reshape(-1), group).reshape(q.shape)
        return q

    @property
    def local_total_charge(self) -> torch.Tensor:
        """The charge of THIS rank's particles, inside or outside `sharding.particle_sharded` (no exchange: safe to read on one rank)."""
        return (self.particle_charges * self.survival_probabilities).sum(dim=-1)

    @property
    def num_particles(self) -> int:
        return self.particles.shape[-2]

    def __len__(self) -> int:
        return int(self.num_particles)

    @property
    def num_particles_survived(self) -> torch.Tensor:
        n = self.survival_probabilities.sum(dim=-1)
        if _SHARDING_STACK:
            from .. import sharding

            group = sharding.active_group()
            if group is not None:           # over ALL shards (a collective; keeps this rank's graph)
                n = sharding.sum_over_ranks(n.reshape(-1), group).reshape(n.shape)
        return n

    # reference frame (beam.py:323-341)
    @property
    def relativistic_gamma(self) -> torch.Tensor:
        return self.energy / self.species.mass_eV

    @property
    def relativistic_beta(self) -> torch.Tensor:
        g = self.relativistic_gamma
        return torch.where(g.abs() > 0, (1 - g.square().reciprocal()).clamp_min(0).sqrt(), torch.ones_like(g))

    @property
    def p0c(self) -> torch.Tensor:
        return self.relativistic_beta * self.relativistic_gamma * self.species.mass_eV

    @property
    def energies(self) -> torch.Tensor:
        return self.p * self.p0c.unsqueeze(-1) + self.energy.unsqueeze(-1)

    @property
    def momenta(self) -> torch.Tensor:
        return (self.energies.square() - self.species.mass_eV.square()).sqrt()

    # derived optics (beam.py:431-556)
    @property
    def emittance_x(self) -> torch.Tensor:
        sp2 = self.sigma_p.square()
        v = ((self.sigma_x.square() - self.cov_xp.square() / sp2) * (self.sigma_px.square() - self.cov_pxp.square() / sp2)
             - (self.cov_xpx - self.cov_xp * self.cov_pxp / sp2).square())
        return v.clamp_min(torch.finfo(v.dtype).tiny).sqrt()

    @property
    def emittance_y(self) -> torch.Tensor:
        sp2 = self.sigma_p.square()
        v = ((self.sigma_y.square() - self.cov_yp.square() / sp2) * (self.sigma_py.square() - self.cov_pyp.square() / sp2)
             - (self.cov_ypy - self.cov_yp * self.cov_pyp / sp2).square())
        return v.clamp_min(torch.finfo(v.dtype).tiny).sqrt()

    @property
    def normalized_emittance_x(self) -> torch.Tensor:
        return self.emittance_x * self.relativistic_beta * self.relativistic_gamma

    @property
    def normalized_emittance_y(self) -> torch.Tensor:
        return self.emittance_y * self.relativistic_beta * self.relativistic_gamma

    @property
    def projected_emittance_x(self) -> torch.Tensor:
        return (self.sigma_x.square() * self.sigma_px.square() - self.cov_xpx.square()).sqrt()

    @property
    def projected_emittance_y(self) -> torch.Tensor:
        return (self.sigma_y.square() * self.sigma_py.square() - self.cov_ypy.square()).sqrt()

    @property
    def beta_x(self) -> torch.Tensor:
        return (self.sigma_x.square() - self.cov_xp.square() / self.sigma_p.square()) / self.emittance_x

    @property
    def beta_y(self) -> torch.Tensor:
        return (self.sigma_y.square() - self.cov_yp.square() / self.sigma_p.square()) / self.emittance_y

    @property
    def alpha_x(self) -> torch.Tensor:
        return -(self.cov_xpx - self.cov_xp * self.cov_pxp / self.sigma_p.square()) / self.emittance_x

    @property
    def alpha_y(self) -> torch.Tensor:
        return -(self.cov_ypy - self.cov_yp * self.cov_pyp / self.sigma_p.square()) / self.emittance_y

    @property
    def dispersion_x(self) -> torch.Tensor:
        return self.cov_xp / self.sigma_p.square()

    @property
    def dispersion_px(self) -> torch.Tensor:
        return self.cov_pxp / self.sigma_p.square()

    @property
    def dispersion_y(self) -> torch.Tensor:
        return self.cov_yp / self.sigma_p.square()

    @property
    def dispersion_py(self) -> torch.Tensor:
        return self.cov_pyp / self.sigma_p.square()

    # ------------------------------------------------------------------ derived beams (host-side plumbing)
    @classmethod
    def make_linspaced(cls, num_particles=10, mu_x=None, mu_px=None, mu_y=None, mu_py=None, mu_tau=None, mu_p=None,
                       sigma_x=None, sigma_px=None, sigma_y=None, sigma_py=None, sigma_tau=None, sigma_p=None,
                       energy=None, total_charge=None, s=None, species=None, device=None, dtype=None) -> "ParticleBeam":
        """`num_particles` particles spaced evenly over mu ± sigma in every coordinate (particle_beam.py:668-800)."""
        fk = {"device": device, "dtype": dtype}
        species = species if species is not None else Species("electron", **fk)
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        mus = [d(mu_x, 0.0), d(mu_px, 0.0), d(mu_y, 0.0), d(mu_py, 0.0), d(mu_tau, 0.0), d(mu_p, 0.0)]
        sigmas = [d(sigma_x, 175e-9), d(sigma_px, 2e-7), d(sigma_y, 175e-9), d(sigma_py, 2e-7), d(sigma_tau, 1e-6),
                  d(sigma_p, 1e-6)]
        energy = d(energy, 1e8)
        total_charge = total_charge if total_charge is not None else species.charge_coulomb * num_particles
        charges = torch.ones((*total_charge.shape, num_particles), **fk) * total_charge.unsqueeze(-1) / num_particles
        vector_shape = torch.broadcast_shapes(*[t.shape for t in mus + sigmas])
        particles = torch.ones((*vector_shape, num_particles, 7), **fk)
        ramp = torch.linspace(0.0, 1.0, num_particles, **fk)
        for i, (mu, sigma) in enumerate(zip(mus, sigmas)):
            lo, hi = (mu - sigma).expand(vector_shape).unsqueeze(-1), (mu + sigma).expand(vector_shape).unsqueeze(-1)
            particles[..., i] = lo + (hi - lo) * ramp
        return cls(particles=particles, energy=energy, particle_charges=charges, s=s, species=species, **fk)

    def linspaced(self, num_particles: int) -> "ParticleBeam":
        """Evenly spaced beam with this beam's means, sigmas, energy and total charge (particle_beam.py:1180-1210)."""
        return self.make_linspaced(
            num_particles=num_particles, mu_x=self.mu_x, mu_px=self.mu_px, mu_y=self.mu_y, mu_py=self.mu_py,
            mu_tau=self.mu_tau, mu_p=self.mu_p, sigma_x=self.sigma_x, sigma_px=self.sigma_px, sigma_y=self.sigma_y,
            sigma_py=self.sigma_py, sigma_tau=self.sigma_tau, sigma_p=self.sigma_p, energy=self.energy,
            total_charge=self.total_charge, s=self.s, species=self.species, device=self.particles.device,
            dtype=self.particles.dtype)

    def randomly_subsampled(self, num_particles: int, adjust_particle_charges: bool = True,
                            random_state=None) -> "ParticleBeam":
        """Random subset of the particles, charges rescaled to keep the total charge (particle_beam.py:1212-1262)."""
        assert num_particles <= self.num_particles, (
            "Number of particles to sample must be less than or equal to the number of particles in the original beam.")
        idx = torch.randperm(self.num_particles, generator=random_state, device=self.particles.device)[:num_particles]
        sub = self.__class__(particles=self.particles[..., idx, :], energy=self.energy,
                             particle_charges=self.particle_charges[..., idx],
                             survival_probabilities=self.survival_probabilities[..., idx], species=self.species)
        if adjust_particle_charges:
            sub.particle_charges = sub.particle_charges * (self.total_charge / sub.total_charge).unsqueeze(-1)
        return sub

    def transformed_to(self, mu_x=None, mu_px=None, mu_y=None, mu_py=None, mu_tau=None, mu_p=None, sigma_x=None,
                       sigma_px=None, sigma_y=None, sigma_py=None, sigma_tau=None, sigma_p=None, energy=None,
                       total_charge=None, species=None) -> "ParticleBeam":
        """Shift and scale every coordinate to new means / sigmas (particle_beam.py:1034-1158); the current moments
        come from one fused `chx_moments` call."""
        names = ("x", "px", "y", "py", "tau", "p")
        given_mu = (mu_x, mu_px, mu_y, mu_py, mu_tau, mu_p)
        given_sigma = (sigma_x, sigma_px, sigma_y, sigma_py, sigma_tau, sigma_p)
        old_mu = torch.stack(torch.broadcast_tensors(*[getattr(self, f"mu_{n}") for n in names]), dim=-1)
        old_sigma = torch.stack(torch.broadcast_tensors(*[getattr(self, f"sigma_{n}") for n in names]), dim=-1)
        new_mu = torch.stack(torch.broadcast_tensors(
            *[g if g is not None else getattr(self, f"mu_{n}") for g, n in zip(given_mu, names)]), dim=-1)
        new_sigma = torch.stack(torch.broadcast_tensors(
            *[g if g is not None else getattr(self, f"sigma_{n}") for g, n in zip(given_sigma, names)]), dim=-1)
        if total_charge is None:
            charges = self.particle_charges
        else:
            charges = (torch.ones_like(self.particle_charges) * total_charge.unsqueeze(-1)
                       / self.particle_charges.shape[-1])
        phase_space = ((self.particles[..., :6] - old_mu.unsqueeze(-2)) / old_sigma.unsqueeze(-2)
                       * new_sigma.unsqueeze(-2) + new_mu.unsqueeze(-2))
        particles = torch.cat([phase_space, torch.ones_like(phase_space[..., :1])], dim=-1)
        return self.__class__(particles=particles, energy=energy if energy is not None else self.energy,
                              particle_charges=charges, survival_probabilities=self.survival_probabilities, s=self.s,
                              species=species if species is not None else self.species)

    # ------------------------------------------------------------------ housekeeping
    @property
    def defining_features(self) -> list[str]:
        """particle_beam.py `defining_features`: what makes two beams equal / what `.to()` converts."""
        return ["particles", "energy", "particle_charges", "survival_probabilities", "s", "species"]

    def clone(self) -> "ParticleBeam":
        """particle_beam.py `clone`: every tensor of the beam and of its species copied — by ONE launch (`clone_many`) where no
        gradient has to flow through the copy."""
        sp = self.species
        p, e, q, w, s, nq, m = _ops.clone_many((self.particles, self.energy, self.particle_charges, self.survival_probabilities,
                                                self.s, sp.num_elementary_charges, sp.mass_eV))
        return self.__class__(particles=p, energy=e, particle_charges=q, survival_probabilities=w, s=s,
                              species=sp._from_tensors(sp.name, nq, m))

    def _snapshot(self) -> "ParticleBeam":
        """Copy of the tensor state (autograd-connected) sharing the species object: what a Screen records, so that
        later in-place edits of this beam do not reach the reading. Cheaper than `clone()` (no new Species tensors)."""
        p, e, q, w, s = _ops.clone_many((self.particles, self.energy, self.particle_charges, self.survival_probabilities, self.s))
        return self.__class__(particles=p, energy=e, particle_charges=q, survival_probabilities=w, s=s, species=self.species)

    def _view(self) -> "ParticleBeam":
        """New beam object sharing this beam's tensors (zero-copy). Pass-through elements return this
        instead of the reference's deep `clone()` (marker.py:52-53): beams are treated as immutable
        values on the tracking path, so no 56-byte-per-particle copy is made for an identity map."""
        return self.__class__(particles=self.particles, energy=self.energy, particle_charges=self.particle_charges,
                              survival_probabilities=self.survival_probabilities, s=self.s, species=self.species)

    def __getitem__(self, item) -> "ParticleBeam":
        vs = torch.broadcast_shapes(self.particles.shape[:-2], self.energy.shape, self.particle_charges.shape[:-1],
                                    self.survival_probabilities.shape[:-1])
        n = self.num_particles
        return self.__class__(
            particles=torch.broadcast_to(self.particles, (*vs, n, 7))[item],
            energy=torch.broadcast_to(self.energy, vs)[item],
            particle_charges=torch.broadcast_to(self.particle_charges, (*vs, n))[item],
            survival_probabilities=torch.broadcast_to(self.survival_probabilities, (*vs, n))[item],
            species=self.species)

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(particles={self.particles}, energy={self.energy}, "
                f"particle_charges={self.particle_charges}, "
                f"survival_probabilities={self.survival_probabilities}, s={self.s}, species={self.species!r})")


def _install_coordinate_properties() -> None:
    for i, name in enumerate(_COORDS):
        def getter(self, i=i):
            return self.particles[..., i]

        def setter(self, value, i=i):
            self.particles[..., i] = value

        setattr(ParticleBeam, name, property(getter, setter))
        setattr(ParticleBeam, f"mu_{name}", property(lambda self, i=i: self._mu(i)))
        setattr(ParticleBeam, f"sigma_{name}", property(lambda self, i=i: self._sigma(i)))
    for i in range(6):
        for j in range(i + 1, 6):
            setattr(ParticleBeam, f"cov_{_COORDS[i]}{_COORDS[j]}", property(lambda self, i=i, j=j: self._cov(i, j)))


def _install_buffer_accessors() -> None:
    """Class-level data descriptors for the five state tensors and the species: attribute reads hit the
    descriptor (one dict lookup) instead of falling through to nn.Module.__getattr__; writes still go
    through nn.Module.__setattr__, which stores tensors into `_buffers`."""
    for name in ("particles", "energy", "particle_charges", "survival_probabilities", "s"):
        def getter(self, name=name):
            v = self._buffers.get(name)
            if v is None:
                v = self._parameters.get(name)
                if v is None:     # nn.Module.__setattr__ probes with hasattr() while it moves a name between the dictionaries
                    raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")
            return v

        def setter(self, value, name=name):
            self._parameters.pop(name, None)
            self._buffers[name] = value

        setattr(ParticleBeam, name, property(getter, setter))

    def get_species(self):
        return self._modules["species"]

    def set_species(self, value):
        self._modules["species"] = value

    ParticleBeam.species = property(get_species, set_species)


_install_buffer_accessors()
_install_coordinate_properties()

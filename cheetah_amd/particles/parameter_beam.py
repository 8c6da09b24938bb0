"""ParameterBeam (mirror of cheetah/particles/parameter_beam.py:8-779): a beam described by its 7-vector of means
`mu (…, 7)` (last entry 1) and its covariance matrix `cov (…, 7, 7)` (last row / column 0).

Tracking is `mu' = R mu`, `cov' = R cov R^T` through the `chx_parameter_track` kernel
(cheetah/accelerator/element.py:167-179); this is the representation most RL environments track.
"""

from __future__ import annotations

import torch

from .. import _ops
from .beam import Beam
from .species import Species

_COORDS = ["x", "px", "y", "py", "tau", "p"]


class ParameterBeam(Beam):
    """Gaussian beam given by `mu` and `cov`."""

    UNVECTORIZED_NUM_ATTR_DIMS = Beam.UNVECTORIZED_NUM_ATTR_DIMS | {"mu": 1, "cov": 2}

    def __init__(self, mu, cov, energy, total_charge=None, s=None, species=None, device=None, dtype=None) -> None:
        super().__init__()
        device = device if device is not None else mu.device
        dtype = dtype if dtype is not None else mu.dtype
        fk = {"device": device, "dtype": dtype}
        assert mu.shape[-1] == 7 and cov.shape[-2:] == (7, 7), "mu must be (…, 7) and cov (…, 7, 7)"
        self._modules["species"] = species if species is not None else Species("electron", **fk)
        # a new beam object per tracked run and per diagnostic: the state tensors are entered into the module's dictionaries
        # directly (same result as five `register_buffer` / `register_parameter` calls, ~15 us cheaper) and read back through
        # the class-level properties installed below instead of nn.Module.__getattr__
        buffers, parameters = self._buffers, self._parameters
        for name, value in (("mu", mu), ("cov", cov), ("energy", energy),
                            ("total_charge", total_charge if total_charge is not None else torch.tensor(0.0, **fk)),
                            ("s", s if s is not None else torch.tensor(0.0, **fk))):
            if isinstance(value, torch.nn.Parameter):
                parameters[name] = value
            else:
                buffers[name] = value

    # ------------------------------------------------------------------ factories
    @classmethod
    def from_parameters(cls, mu_x=None, mu_px=None, mu_y=None, mu_py=None, mu_tau=None, mu_p=None, sigma_x=None,
                        sigma_px=None, sigma_y=None, sigma_py=None, sigma_tau=None, sigma_p=None, cov_xpx=None,
                        cov_ypy=None, cov_taup=None, cov_xp=None, cov_pxp=None, cov_yp=None, cov_pyp=None,
                        cov_xy=None, cov_xpy=None, cov_xtau=None, cov_pxy=None, cov_pxpy=None, cov_pxtau=None,
                        cov_ytau=None, cov_pytau=None, energy=None, total_charge=None, s=None, species=None,
                        device=None, dtype=None) -> "ParameterBeam":
        """parameter_beam.py:61-280 (same defaults)."""
        fk = {"device": device, "dtype": dtype}
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        mus = torch.broadcast_tensors(d(mu_x, 0.0), d(mu_px, 0.0), d(mu_y, 0.0), d(mu_py, 0.0), d(mu_tau, 0.0),
                                      d(mu_p, 0.0))
        mu = torch.stack([*mus, torch.ones_like(mus[0])], dim=-1)
        entries = {
            (0, 0): d(sigma_x, 175e-6).square(), (1, 1): d(sigma_px, 4e-6).square(), (2, 2): d(sigma_y, 175e-6).square(),
            (3, 3): d(sigma_py, 4e-6).square(), (4, 4): d(sigma_tau, 8e-6).square(), (5, 5): d(sigma_p, 2e-3).square(),
            (0, 1): d(cov_xpx, 0.0), (2, 3): d(cov_ypy, 0.0), (4, 5): d(cov_taup, 0.0), (0, 5): d(cov_xp, 0.0),
            (1, 5): d(cov_pxp, 0.0), (2, 5): d(cov_yp, 0.0), (3, 5): d(cov_pyp, 0.0), (0, 2): d(cov_xy, 0.0),
            (0, 3): d(cov_xpy, 0.0), (0, 4): d(cov_xtau, 0.0), (1, 2): d(cov_pxy, 0.0), (1, 3): d(cov_pxpy, 0.0),
            (1, 4): d(cov_pxtau, 0.0), (2, 4): d(cov_ytau, 0.0), (3, 4): d(cov_pytau, 0.0),
        }
        shape = torch.broadcast_shapes(*[v.shape for v in entries.values()])
        cov = torch.zeros(*shape, 7, 7, **fk)
        for (i, j), v in entries.items():
            cov[..., i, j] = v
            cov[..., j, i] = v
        try:
            torch.linalg.cholesky(cov[..., :6, :6])
        except RuntimeError as e:
            raise ValueError("The covariance matrix of the beam must be positive definite. Please check the input "
                             "parameters to ensure that they are consistent.") from e
        energy = energy if energy is not None else torch.tensor(1e8, **fk)
        return cls(mu=mu, cov=cov, energy=energy, total_charge=total_charge, s=s, species=species, device=device,
                   dtype=dtype)

    @classmethod
    def from_twiss(cls, beta_x=None, alpha_x=None, emittance_x=None, beta_y=None, alpha_y=None, emittance_y=None,
                   sigma_tau=None, sigma_p=None, cov_taup=None, dispersion_x=None, dispersion_px=None, dispersion_y=None,
                   dispersion_py=None, energy=None, total_charge=None, s=None, species=None, device=None,
                   dtype=None) -> "ParameterBeam":
        """Twiss parameters and dispersion -> moments (parameter_beam.py:282-414)."""
        fk = {"device": device, "dtype": dtype}
        d = lambda v, default: v if v is not None else torch.tensor(default, **fk)  # noqa: E731
        beta_x, alpha_x, emittance_x = d(beta_x, 1.0), d(alpha_x, 0.0), d(emittance_x, 7.1971891e-13)
        beta_y, alpha_y, emittance_y = d(beta_y, 1.0), d(alpha_y, 0.0), d(emittance_y, 7.1971891e-13)
        dx, dpx, dy, dpy = d(dispersion_x, 0.0), d(dispersion_px, 0.0), d(dispersion_y, 0.0), d(dispersion_py, 0.0)
        sigma_p = d(sigma_p, 1e-6)
        sp2 = sigma_p.square()
        return cls.from_parameters(
            sigma_x=(emittance_x * beta_x + dx.square() * sp2).sqrt(),
            sigma_px=(emittance_x * (1 + alpha_x.square()) / beta_x + dpx.square() * sp2).sqrt(),
            sigma_y=(emittance_y * beta_y + dy.square() * sp2).sqrt(),
            sigma_py=(emittance_y * (1 + alpha_y.square()) / beta_y + dpy.square() * sp2).sqrt(),
            sigma_tau=d(sigma_tau, 1e-6), sigma_p=sigma_p, cov_xpx=-emittance_x * alpha_x + dx * dpx * sp2,
            cov_ypy=-emittance_y * alpha_y + dy * dpy * sp2, cov_taup=d(cov_taup, 0.0), cov_xp=dx * sp2, cov_pxp=dpx * sp2,
            cov_yp=dy * sp2, cov_pyp=dpy * sp2, energy=d(energy, 1e8), total_charge=total_charge, s=s, species=species,
            device=device, dtype=dtype)
    @classmethod
    def from_ocelot(cls, parray, device=None, dtype=None) -> "ParameterBeam":
        """Moments of an Ocelot `ParticleArray` (parameter_beam.py:416-442)."""
        kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
        mu = torch.ones(7, **kw)
        mu[:6] = torch.as_tensor(parray.rparticles.mean(axis=1), **kw)
        cov = torch.zeros(7, 7, **kw)
        cov[:6, :6] = torch.as_tensor(parray.rparticles, **kw).cov()
        return cls(mu=mu, cov=cov, energy=1e9 * torch.as_tensor(parray.E, **kw),
                   total_charge=torch.as_tensor(parray.q_array, **kw).sum(), species=Species("electron", **kw), **kw)

    def linspaced(self, num_particles: int):
        """ParticleBeam with `num_particles` evenly spaced particles and this beam's parameters
        (parameter_beam.py:566-600)."""
        from .particle_beam import ParticleBeam

        return ParticleBeam.make_linspaced(
            num_particles=num_particles, mu_x=self.mu_x, mu_y=self.mu_y, mu_px=self.mu_px, mu_py=self.mu_py,
            sigma_x=self.sigma_x, sigma_y=self.sigma_y, sigma_px=self.sigma_px, sigma_py=self.sigma_py,
            sigma_tau=self.sigma_tau, sigma_p=self.sigma_p, energy=self.energy, total_charge=self.total_charge,
            species=self.species, device=self.mu.device, dtype=self.mu.dtype)

    @classmethod
    def from_astra(cls, path: str, device=None, dtype=None) -> "ParameterBeam":
        """Moments of an Astra particle distribution (parameter_beam.py:444-474, converters/astra.py)."""
        from ..converters.astra import from_astrabeam

        kw = {"device": device or torch.get_default_device(), "dtype": dtype or torch.get_default_dtype()}
        coordinates, energy, charges = from_astrabeam(path)
        mu = torch.ones(7, **kw)
        mu[:6] = torch.as_tensor(coordinates.mean(axis=0), **kw)
        cov = torch.zeros(7, 7, **kw)
        cov[:6, :6] = torch.as_tensor(coordinates, **kw).T.cov()
        return cls(mu=mu, cov=cov, energy=torch.as_tensor(energy, **kw), total_charge=torch.as_tensor(charges, **kw).sum(),
                   species=Species("electron"), **kw)


    @property
    def defining_features(self) -> list[str]:
        return ["mu", "cov", "energy", "total_charge", "s", "species"]

    def transformed_to(self, mu_x=None, mu_px=None, mu_y=None, mu_py=None, mu_tau=None, mu_p=None, sigma_x=None,
                       sigma_px=None, sigma_y=None, sigma_py=None, sigma_tau=None, sigma_p=None, cov_xpx=None, cov_ypy=None, cov_taup=None, cov_xp=None, cov_pxp=None, cov_yp=None, cov_pyp=None, cov_xy=None, cov_xpy=None, cov_xtau=None,
                       cov_pxy=None, cov_pxpy=None, cov_pxtau=None, cov_ytau=None, cov_pytau=None,
                       energy=None, total_charge=None, species=None, device=None, dtype=None) -> "ParameterBeam":
        """New beam with some moments replaced (parameter_beam.py:476-586): unspecified mu_* / sigma_* / cov_* keep
        their current value."""
        given = dict(locals())
        moments = {k: v for k, v in given.items() if k.startswith(("mu_", "sigma_", "cov_"))}
        current = {f"mu_{c}": getattr(self, f"mu_{c}") for c in _COORDS}
        current.update({f"sigma_{c}": getattr(self, f"sigma_{c}") for c in _COORDS})
        for i in range(6):
            for j in range(i + 1, 6):
                name = f"cov_{_COORDS[i]}{_COORDS[j]}"
                current[name] = getattr(self, name)
        unknown = set(moments) - set(current)
        if unknown:
            raise TypeError(f"unknown beam parameters {sorted(unknown)}")
        current.update({k: v for k, v in moments.items() if v is not None})
        return self.__class__.from_parameters(
            **current, energy=energy if energy is not None else self.energy,
            total_charge=total_charge if total_charge is not None else self.total_charge, s=self.s,
            species=species if species is not None else self.species,
            device=device if device is not None else self.mu.device, dtype=dtype if dtype is not None else self.mu.dtype)

    def as_particle_beam(self, num_particles: int):
        """parameter_beam.py:588-608."""
        from .particle_beam import ParticleBeam

        return ParticleBeam.from_distribution(num_particles=num_particles, mu=self.mu[..., :6], cov=self.cov[..., :6, :6],
                                              energy=self.energy, total_charge=self.total_charge, s=self.s,
                                              species=self.species, device=self.mu.device, dtype=self.mu.dtype)

    # ------------------------------------------------------------------ tracking primitive
    def _tracked(self, tm: torch.Tensor, length, cavity_coeffs=None, energy=None, batch_shape=None) -> "ParameterBeam":
        mu, cov = _ops.parameter_track(self.mu, self.cov, tm, cavity_coeffs, batch_shape)
        return self.__class__(mu, cov, self.energy if energy is None else energy, total_charge=self.total_charge,
                              s=self.s + length if length is not None else self.s, species=self.species)

    @classmethod
    def _from_moment_vector(cls, mom: torch.Tensor, dtype, energy, total_charge=None, s=None, species=None):
        """(…,29) [W, W2, mu(6), cov upper triangle(21)] (chx_moments layout) -> ParameterBeam with the affine 7th
        coordinate (mu_6 = 1, zero covariance), like particle_beam.py `as_parameter_beam`."""
        iu = torch.triu_indices(6, 6, device=mom.device)
        cov = torch.zeros((*mom.shape[:-1], 7, 7), dtype=mom.dtype, device=mom.device)
        cov[..., iu[0], iu[1]] = mom[..., 8:29]
        cov[..., iu[1], iu[0]] = mom[..., 8:29]
        mu = torch.cat([mom[..., 2:8], torch.ones_like(mom[..., :1])], dim=-1)
        return cls(mu.to(dtype), cov.to(dtype), energy, total_charge=total_charge, s=s, species=species)

    def _snapshot(self) -> "ParameterBeam":
        mu, cov, energy, q, s = _ops.clone_many((self.mu, self.cov, self.energy, self.total_charge, self.s))
        return self.__class__(mu, cov, energy, total_charge=q, s=s, species=self.species)

    def _view(self) -> "ParameterBeam":
        return self.__class__(self.mu, self.cov, self.energy, total_charge=self.total_charge, s=self.s,
                              species=self.species)

    def clone(self) -> "ParameterBeam":
        sp = self.species
        mu, cov, energy, q, s, nq, m = _ops.clone_many((self.mu, self.cov, self.energy, self.total_charge, self.s,
                                                        sp.num_elementary_charges, sp.mass_eV))
        return self.__class__(mu=mu, cov=cov, energy=energy, total_charge=q, s=s, species=sp._from_tensors(sp.name, nq, m))

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(mu={self.mu!r}, cov={self.cov!r}, energy={self.energy!r}, "
                f"total_charge={self.total_charge!r}, s={self.s!r}, species={self.species!r})")


def _install_moment_properties() -> None:
    for i, name in enumerate(_COORDS):
        setattr(ParameterBeam, f"mu_{name}", property(lambda self, i=i: self.mu[..., i]))
        setattr(ParameterBeam, f"sigma_{name}", property(lambda self, i=i: self.cov[..., i, i].clamp_min(0).sqrt()))
    for i in range(6):
        for j in range(i + 1, 6):
            setattr(ParameterBeam, f"cov_{_COORDS[i]}{_COORDS[j]}", property(lambda self, i=i, j=j: self.cov[..., i, j]))


_install_moment_properties()


def _install_state_accessors() -> None:
    """Class-level data descriptors for the five state tensors and the species (see ParticleBeam): reads are one or two
    dictionary lookups instead of nn.Module.__getattr__; a tensor assigned later lands where nn.Module would put it."""
    for name in ("mu", "cov", "energy", "total_charge", "s"):
        def getter(self, name=name):
            v = self._buffers.get(name)
            if v is None:
                v = self._parameters.get(name)
                if v is None:     # nn.Module.__setattr__ probes with hasattr() while it moves a name between the dictionaries
                    raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")
            return v

        def setter(self, value, name=name):
            if isinstance(value, torch.nn.Parameter):
                self._buffers.pop(name, None)
                self._parameters[name] = value
            else:
                self._parameters.pop(name, None)
                self._buffers[name] = value

        setattr(ParameterBeam, name, property(getter, setter))

    def get_species(self):
        return self._modules["species"]

    def set_species(self, value):
        self._modules["species"] = value

    ParameterBeam.species = property(get_species, set_species)


_install_state_accessors()

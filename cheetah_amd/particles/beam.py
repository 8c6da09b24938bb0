"""Beam base class: reference-frame quantities and derived optics shared by ParticleBeam and ParameterBeam
(mirror of cheetah/particles/beam.py:323-556). Everything here is scalar post-processing of the first and
second moments, which the subclasses provide (`mu_*`, `sigma_*`, `cov_*`)."""

from __future__ import annotations

import torch
from torch import nn


_OUT_OF_SCOPE = ("{} is plotting / visualisation, outside this tracking engine (SURVEY.md section 2): move the data to the host "
                 "(`.cpu()`) and plot it there, or hand it to the reference's plotting helpers")


def _abstract(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.{name} is provided by ParticleBeam and ParameterBeam")

    method.__name__ = name
    return method


class Beam(nn.Module):
    """Common interface of the two beam representations (beam.py:17-592: the abstract surface is declared here too, so that
    `hasattr(Beam, ...)` and subclass checks of user code see the same names)."""

    UNVECTORIZED_NUM_ATTR_DIMS: dict = {}          # beam.py:36: dimensions of each tensor attribute without vector dims

    def register_buffer_or_parameter(self, name: str, value) -> None:
        """beam.py:574-589"""
        if isinstance(value, nn.Parameter):
            self.register_parameter(name, value)
        else:
            self.register_buffer(name, value)

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


# the abstract part of the reference's interface (beam.py:38-321, 343-430): implemented by the two subclasses
for _name in ("from_parameters", "from_twiss", "from_astra", "from_ocelot"):
    setattr(Beam, _name, classmethod(_abstract(_name)))
for _name in ("transformed_to", "clone"):
    setattr(Beam, _name, _abstract(_name))
for _name in (["defining_features"] + [f"mu_{c}" for c in ("x", "px", "y", "py", "tau", "p")]
              + [f"sigma_{c}" for c in ("x", "px", "y", "py", "tau", "p")]
              + ["cov_xpx", "cov_ypy", "cov_taup", "cov_xp", "cov_pxp", "cov_yp", "cov_pyp", "cov_xy", "cov_xpy", "cov_xtau",
                 "cov_pxy", "cov_pxpy", "cov_pxtau", "cov_ytau", "cov_pytau"]):
    setattr(Beam, _name, property(_abstract(_name)))
del _name

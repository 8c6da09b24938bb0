"""Beam base class: reference-frame quantities and derived optics shared by ParticleBeam and ParameterBeam
(mirror of cheetah/particles/beam.py:323-556). Everything here is scalar post-processing of the first and
second moments, which the subclasses provide (`mu_*`, `sigma_*`, `cov_*`)."""

from __future__ import annotations

import torch
from torch import nn


class Beam(nn.Module):
    """Common interface of the two beam representations."""

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

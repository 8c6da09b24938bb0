"""Particle species (mirror of cheetah/particles/species.py:12-149).

Mass and charge are kept as tensors for API compatibility and additionally cached as Python floats
(`mass_eV_float`, `num_elementary_charges_float`): the libchx entry points take them by value, so no
device->host synchronisation happens on the tracking path.
"""

from __future__ import annotations

import torch
from torch import nn

# scipy.constants / CODATA 2022 values used by the reference (species.py:5-9, scipy 1.15.3)
electron_mass_eV = 510998.95069
proton_mass_eV = 938272089.4300001
deuteron_mass_eV = 1875612945.0
elementary_charge = 1.602176634e-19
eV_to_kg = 1.7826619216278975e-36


class Species(nn.Module):
    """Named particle species defined by charge and mass."""

    def register_buffer_or_parameter(self, name: str, value) -> None:
        """species.py:118-131"""
        if isinstance(value, nn.Parameter):
            self.register_parameter(name, value)
        else:
            self.register_buffer(name, value)

    known = {
        "electron": {"num_elementary_charges": -1, "mass_eV": electron_mass_eV},
        "positron": {"num_elementary_charges": 1, "mass_eV": electron_mass_eV},
        "proton": {"num_elementary_charges": 1, "mass_eV": proton_mass_eV},
        "antiproton": {"num_elementary_charges": -1, "mass_eV": proton_mass_eV},
        "deuteron": {"num_elementary_charges": 1, "mass_eV": deuteron_mass_eV},
    }

    def __init__(self, name, num_elementary_charges=None, charge_coulomb=None, mass_eV=None, mass_kg=None,
                 device=None, dtype=None) -> None:
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.name = name
        if name in self.known:
            assert all(v is None for v in (num_elementary_charges, charge_coulomb, mass_eV, mass_kg)), \
                "Known particle species should not have charge and mass provided."
            nq = torch.tensor(self.known[name]["num_elementary_charges"], **factory_kwargs)
            m = torch.tensor(self.known[name]["mass_eV"], **factory_kwargs)
        else:
            assert (num_elementary_charges is not None or charge_coulomb is not None) and (
                mass_eV is not None or mass_kg is not None
            ), "Custom particle species should have charge and mass provided."
            assert not (num_elementary_charges is not None and charge_coulomb is not None), \
                "Only one of charge_elementary and charge_coulomb should be provided."
            assert not (mass_eV is not None and mass_kg is not None), \
                "Only one of mass_eV and mass_kg should be provided."
            nq = num_elementary_charges if num_elementary_charges is not None else charge_coulomb / elementary_charge
            m = mass_eV if mass_eV is not None else mass_kg / eV_to_kg
        self._register("num_elementary_charges", nq)
        self._register("mass_eV", m)

    def _register(self, name, value):
        if isinstance(value, nn.Parameter):
            self.register_parameter(name, value)
        else:
            self.register_buffer(name, value)

    # ---- host-side scalars for the C-ABI (refreshed when the tensors are replaced / moved) ----
    def _scalars(self):
        buffers, params = self._buffers, self._parameters
        m = buffers["mass_eV"] if "mass_eV" in buffers else params["mass_eV"]
        q = buffers["num_elementary_charges"] if "num_elementary_charges" in buffers else params["num_elementary_charges"]
        cached = self.__dict__.get("_scalar_cache")
        if cached is None or cached[0][0] is not m or cached[0][1] != m._version or cached[0][2] is not q \
                or cached[0][3] != q._version:
            cached = ((m, m._version, q, q._version), float(m.detach().double().reshape(-1)[0].item()),
                      float(q.detach().double().reshape(-1)[0].item()))
            self.__dict__["_scalar_cache"] = cached
        return cached[1], cached[2]

    @property
    def mass_eV_float(self) -> float:
        return self._scalars()[0]

    @property
    def num_elementary_charges_float(self) -> float:
        return self._scalars()[1]

    @property
    def mass_kg(self) -> torch.Tensor:
        return self.mass_eV * eV_to_kg

    @property
    def charge_coulomb(self) -> torch.Tensor:
        return self.num_elementary_charges * elementary_charge

    def clone(self) -> "Species":
        """species.py:133-149. Charge and mass are copied on their device (the reference re-creates a known species from its
        constants, which on a GPU would be two synchronous host-to-device transfers per cloned beam; the values are the same)."""
        from .. import _ops

        nq, m = _ops.clone_many((self.num_elementary_charges, self.mass_eV))
        return self._from_tensors(self.name, nq, m)

    @classmethod
    def _from_tensors(cls, name: str, num_elementary_charges: torch.Tensor, mass_eV: torch.Tensor) -> "Species":
        sp = cls.__new__(cls)
        nn.Module.__init__(sp)
        sp.name = name
        sp._register("num_elementary_charges", num_elementary_charges)
        sp._register("mass_eV", mass_eV)
        return sp

    def __repr__(self) -> str:
        return (f"Species(name={self.name!r}, num_elementary_charges={self.num_elementary_charges!r}, "
                f"mass_eV={self.mass_eV!r})")

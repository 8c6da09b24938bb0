"""Import of foreign lattice and beam formats (mirror of cheetah/converters): Bmad and Elegant lattice files, the ARES
NX-tables export, Ocelot cells (duck-typed, Ocelot itself is not needed) and Astra particle distributions. Host-side
parsing only: the elements these produce track through the same libchx kernels as hand-built lattices."""

from . import astra, bmad, elegant, nxtables, ocelot  # noqa: F401

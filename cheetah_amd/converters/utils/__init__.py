"""Expression evaluators under the reference's module names (cheetah/converters/utils/{infix,rpn}.py). The lattice-file
reader itself (the reference's fortran_namelist module) is `cheetah_amd.converters.lattice_text`."""
from . import infix, rpn  # noqa: F401

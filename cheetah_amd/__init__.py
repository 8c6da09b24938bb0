"""cheetah_amd — MI355X-native tracking engine behind Cheetah's Segment / Element / ParticleBeam API.

The hot path `Segment.track(ParticleBeam)` (map builders, linear apply, cavity, beam moments, screen
images, space-charge kick) runs in hand-written HIP kernels for gfx950 exposed through the C-ABI of
`libchx.so` (include/chx.h). There is no CPU or eager fallback: tensors must live on a ROCm device.
"""

from . import _lib  # noqa: F401
from .accelerator import (  # noqa: F401
    BPM,
    Aperture,
    Cavity,
    CombinedCorrector,
    CustomTransferMap,
    Dipole,
    Drift,
    Element,
    HorizontalCorrector,
    Marker,
    PhysicsWarning,
    Quadrupole,
    RBend,
    Screen,
    Segment,
    Sextupole,
    TransverseDeflectingCavity,
    Solenoid,
    SpaceChargeKick,
    Superimposed,
    Undulator,
    VerticalCorrector,
)
from .particles import ParameterBeam, ParticleBeam, Species  # noqa: F401
from . import converters, graph, latticejson, track_methods, utils  # noqa: F401,E402
from .particles.beam import Beam  # noqa: F401,E402
from .warnings import (  # noqa: F401,E402
    DefaultParameterWarning,
    DirtyNameWarning,
    NoBeamPropertiesInLatticeWarning,
    NotUnderstoodPropertyWarning,
    UnknownElementWarning,
    VisualizationWarning,
)

__version__ = "0.1.0"

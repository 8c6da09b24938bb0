"""Mirror of cheetah/utils/warnings.py (the classes live in cheetah_amd/warnings.py)."""
from ..warnings import (  # noqa: F401
    DefaultParameterWarning,
    DirtyNameWarning,
    NoBeamPropertiesInLatticeWarning,
    NotUnderstoodPropertyWarning,
    PhysicsWarning,
    UnknownElementWarning,
    VisualizationWarning,
)

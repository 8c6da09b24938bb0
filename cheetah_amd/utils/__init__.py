"""Helper functions user code imports from `cheetah.utils` (same sub-module layout as cheetah/utils/__init__.py, so
`from cheetah.utils.warnings import PhysicsWarning`-style imports keep working after switching packages). Not mirrored:
`plot`, `assets` (visualisation), `device` (MPS probing) and `cache` — maps are cached per element in
`Element._cached_map`, keyed on the defining features like the reference's decorator."""
from . import autograd, bmadx  # noqa: F401
from .cloud_in_cell import cloud_in_cell_charge_deposition  # noqa: F401
from .elementwise_linspace import elementwise_linspace  # noqa: F401
from .kde import kde_histogram_1d, kde_histogram_2d  # noqa: F401
from .names import UniqueNameGenerator, merge_element_names  # noqa: F401
from .physics import compute_relativistic_factors  # noqa: F401
from .statistics import (  # noqa: F401
    match_distribution_moments,
    unbiased_weighted_covariance,
    unbiased_weighted_covariance_matrix,
    unbiased_weighted_std,
    unbiased_weighted_variance,
)
from .vector import squash_index_for_unavailable_dims  # noqa: F401
from .warnings import (  # noqa: F401
    DefaultParameterWarning,
    DirtyNameWarning,
    NoBeamPropertiesInLatticeWarning,
    NotUnderstoodPropertyWarning,
    PhysicsWarning,
    UnknownElementWarning,
    VisualizationWarning,
)

from .particle_beam import ParticleBeam  # noqa: F401
from .species import Species  # noqa: F401

"""Warning categories of the host layer (mirror of cheetah/utils/warnings.py)."""


class PhysicsWarning(Warning):
    """Something that may make the simulated physics differ from what the user expects."""


class UnknownElementWarning(PhysicsWarning):
    """A foreign-lattice element with no counterpart here was replaced by a stand-in (usually a Drift)."""


class NotUnderstoodPropertyWarning(PhysicsWarning):
    """A property of a foreign-lattice element was ignored."""


class NoBeamPropertiesInLatticeWarning(PhysicsWarning):
    """Beam properties found in a lattice file are dropped: they belong to the Beam classes."""


class DefaultParameterWarning(PhysicsWarning):
    """A parameter was not given and fell back to its default."""


class DirtyNameWarning(Warning):
    """An element name that is not a valid Python identifier (no `segment.<name>` access)."""


class VisualizationWarning(Warning):
    """A plot may not show what the user expects."""

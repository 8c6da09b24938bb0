from .cavity import Cavity  # noqa: F401
from .correctors import CombinedCorrector, HorizontalCorrector, VerticalCorrector  # noqa: F401
from .custom_transfer_map import CustomTransferMap  # noqa: F401
from .dipole import Dipole, RBend  # noqa: F401
from .drift import Drift  # noqa: F401
from .element import Element, PhysicsWarning  # noqa: F401
from .marker import BPM, Aperture, Marker  # noqa: F401
from .misc_elements import Sextupole, Solenoid, TransverseDeflectingCavity, Undulator  # noqa: F401
from .quadrupole import Quadrupole  # noqa: F401
from .screen import Screen  # noqa: F401
from .segment import Segment  # noqa: F401
from .space_charge_kick import SpaceChargeKick  # noqa: F401
from .superimposed import Superimposed  # noqa: F401

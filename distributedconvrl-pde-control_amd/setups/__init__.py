from .ks import KSSetup  # noqa: F401
from .keller_segel import KellerSegelSetup  # noqa: F401
from .fluid import FluidSetup  # noqa: F401
from .keller_segel2d import KellerSegel2DSetup  # noqa: F401

from .disco import MultiDISCO  # noqa: F401
from .dual import DualSVMPC  # noqa: F401

from .disco import MultiDISCO  # noqa: F401

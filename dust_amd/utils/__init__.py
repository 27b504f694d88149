from .spaces import Box  # noqa: F401

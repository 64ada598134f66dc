from . import uwa  # noqa: F401

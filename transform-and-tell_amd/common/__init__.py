from .registrable import Registrable  # noqa: F401

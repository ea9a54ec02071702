from . import ops, layers  # noqa: F401

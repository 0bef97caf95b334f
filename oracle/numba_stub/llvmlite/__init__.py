"""Stub of llvmlite (test infrastructure; see ../numba/__init__.py)."""
from . import ir  # noqa: F401

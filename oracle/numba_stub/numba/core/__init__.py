"""Stub of numba.core (test infrastructure)."""
from . import cgutils  # noqa: F401

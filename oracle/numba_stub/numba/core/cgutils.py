"""Stub of numba.core.cgutils (test infrastructure)."""

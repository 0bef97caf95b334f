"""Stub of numba.np (test infrastructure)."""

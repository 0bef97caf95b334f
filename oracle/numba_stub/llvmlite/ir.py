"""Stub of llvmlite.ir (test infrastructure)."""
